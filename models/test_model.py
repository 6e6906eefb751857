"""Entry point of the reference's CIFAR harness (`python models/test_model.py [options] <architecture> ...`,
/root/reference/models/test_model.py) on the HIP-backed networks: chexpert_amd.cifar."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chexpert_amd.cifar import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
