"""networks/pgandeep of the reference: pgan with len(kernel_spec[phase]) convolutions per block."""
