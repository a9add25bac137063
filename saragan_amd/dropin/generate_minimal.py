"""generate_minimal.py of the reference tree -> saragan_amd.generate_minimal."""
from saragan_amd.generate_minimal import build_parser, main  # noqa: F401

if __name__ == '__main__':
    main(build_parser().parse_args())
