#!/usr/bin/env python3
"""`python main.py --test ...` -- the file name and working-directory contract of the reference's diffusion/main.py
(diffusion/testing_scripts/test.sh:24 runs exactly that from diffusion/, with configs/<task>.yml and --exp ./results/...).
Put the reference's configs/ (and testing_scripts/, if wanted) beside this file and the reference's command lines run unchanged on
the MI355X path; everything else lives in nested_diffusion_amd.main (flag for flag the reference's parser, main.py:16-161)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nested_diffusion_amd.main import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
