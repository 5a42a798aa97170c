"""`python music2dance_amd/phase1/train_wgan-gp.py ...` — the reference's script name;
forwards to music2dance_amd.phase1.train_wgan_gp (a hyphenated file cannot be imported)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from music2dance_amd.phase1.train_wgan_gp import main  # noqa: E402

if __name__ == "__main__":
    main()
