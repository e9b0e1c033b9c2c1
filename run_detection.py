#!/usr/bin/env python3
"""Entry point for sliding-window cough detection on the MI355X path
(counterpart of the reference's run_detection.py -> src.inference.main)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cough_detector_amd.inference import main  # noqa: E402

if __name__ == "__main__":
    main()
