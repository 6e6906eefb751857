#!/usr/bin/env python
"""Entry point with the reference's file name: `python chexpert.py --train --model densenet121 --synthetic 512 ...`"""
from chexpert_amd.cli import main

if __name__ == "__main__":
    main()
