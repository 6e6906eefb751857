#!/usr/bin/env python
"""Entry point with the reference's file name: `python predict.py data.csv predictions.csv --restore_path <checkpoint or folder>`"""
from chexpert_amd.predict import main

if __name__ == "__main__":
    main()
