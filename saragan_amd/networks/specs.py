"""Lookup into the --kernel_spec / --filter_spec tables (`spec[phase][layer]`, main.py:225-232, kernel_spec.json).
The reference repeats this lookup as four helper functions per architecture (pgan/generator.py:4-24,
pgan/discriminator.py:3-23, pgandeep/*.py): a missing phase or layer is a ValueError after a message naming the flag."""


def spec_entry(spec, phase_i, layer_i, what):
    """spec[phase_i][layer_i]; `what` is 'filter' or 'kernel' (which flag to blame)."""
    noun = 'filter count' if what == 'filter' else 'kernel shape'
    if phase_i < 0 or phase_i >= len(spec):
        print(f"Error: no {noun} specified for phase {phase_i}. Please check the file passed to --{what}_spec.")
        raise ValueError(f'no {noun} for phase {phase_i}')
    if layer_i < 0 or layer_i >= len(spec[phase_i]):
        print(f"Error: no {noun} specified for layer {layer_i} in phase {phase_i}. "
              f"Please check the file passed to --{what}_spec.")
        raise ValueError(f'no {noun} for layer {layer_i} in phase {phase_i}')
    return spec[phase_i][layer_i]


def filters(filter_spec, phase_i, layer_i):
    return spec_entry(filter_spec, phase_i, layer_i, 'filter')


def kernels(kernel_spec, phase_i, layer_i):
    return spec_entry(kernel_spec, phase_i, layer_i, 'kernel')
