"""End-to-end rate of the real-data training loop on a generated CheXpert-small folder (JPEG decode -> pinned ring -> H2D -> hipGraph
step), with and without the decoded-image cache: python scratch/e2e_cache.py [n_images]   (each run is a child process: the loader's
workers are forked before the GPU is touched)"""
import json, os, subprocess, sys, tempfile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sys.path.insert(0, '.')
from chexpert_amd.loader import make_jpeg_folder
with tempfile.TemporaryDirectory() as root:
    make_jpeg_folder(root, n=n)
    for gb in ("0", "4"):
        out = os.path.join(root, "run_cache" + gb)
        cmd = [sys.executable, "chexpert.py", "--train", "--data_path", root, "--output_dir", out, "--model", "densenet121", "--resize", "320",
               "--batch_size", "256", "--n_epochs", "4", "--fused_optimizer", "--graph", "--num_workers", "16", "--cache_decoded", gb,
               "--eval_interval", "100000", "--log_interval", "100000"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        rates = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and "images_per_sec" in l]
        print("--cache_decoded %s GB:" % gb, [(d["epoch"], d["images_per_sec"], d.get("decoded_cache_fill")) for d in rates], flush=True)
        if r.returncode != 0:
            print(r.stderr[-1500:])
            sys.exit(r.returncode)
