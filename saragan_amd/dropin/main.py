"""`python main.py pgan <dataset_path> --start_shape ...` as in SURFGAN_3D/main.py:209-439 (normal run)."""
from saragan_amd.main import build_parser, finalize_args, main  # noqa: F401

if __name__ == '__main__':
    main()
