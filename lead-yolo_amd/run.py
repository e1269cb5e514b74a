"""`python -m lead_yolo_amd.run <script.py> [args...]` — run an unmodified reference script
(train.py / detect.py / val.py, cwd = the reference checkout) with the HIP modules injected."""
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m lead_yolo_amd.run <script.py> [args...]")
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    sys.path.insert(0, ".")
    from . import inject
    inject.patch()
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
