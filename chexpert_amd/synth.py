"""Counter-based synthetic tensors (build-owned; SURVEY.md §8d "Synthetic inputs").

Every value is a pure function of (seed, element index): a 64-bit integer mix followed by an
exact integer -> float conversion.  No libm call is involved, so the same tensors can be
re-created bit for bit in this container (golden generation against the reference) and on the
GPU box (parity tests, bench) without shipping them.

Mirrors the reference's input pipeline where it matters:
  * images: uint8 U{0..255} -> /255 -> (x - 0.5330) / 0.0349 -> expanded to 3 identical channels
    (/root/reference/chexpert.py:70-72)
  * targets: {0,1} float labels, Bernoulli(0.3) (U-Ones labels, /root/reference/dataset.py:139-142)
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n pseudo-random uint64 words for counters offset..offset+n-1 under `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        key = _mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x1234567))
        return _mix64(idx * np.uint64(0x9E3779B97F4A7C15) + key)


def uniform01(seed: int, n: int) -> np.ndarray:
    """float64 in [0,1) with 24 significant bits (exactly representable in fp32)."""
    return (hash_u64(seed, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def uniform(seed: int, shape, lo=-1.0, hi=1.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = uniform01(seed, n) * (hi - lo) + lo
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape)).to(dtype)


def symmetric(seed: int, shape, std=1.0, dtype=torch.float32) -> torch.Tensor:
    """zero-mean uniform with the requested standard deviation (a = std*sqrt(3))."""
    a = float(std) * 3.0 ** 0.5
    return uniform(seed, shape, -a, a, dtype)


def xray_u8(seed: int, batch: int, size: int) -> torch.Tensor:
    """(B,1,S,S) uint8 iid U{0..255}."""
    n = batch * size * size
    v = (hash_u64(seed, n) >> np.uint64(56)).astype(np.uint8)
    return torch.from_numpy(v).reshape(batch, 1, size, size)


MEAN, STD = 0.5330, 0.0349  # /root/reference/chexpert.py:71


def normalise(u8: torch.Tensor) -> torch.Tensor:
    """uint8 (B,1,S,S) -> float32 (B,3,S,S): the reference transform chain chexpert.py:70-72."""
    x = u8.float().div(255)
    x = (x - MEAN) / STD
    return x.expand(-1, 3, -1, -1).contiguous()


def xray_batch(seed: int, batch: int, size: int = 320) -> torch.Tensor:
    return normalise(xray_u8(seed, batch, size))


def targets(seed: int, batch: int, n_classes: int, p: float = 0.3) -> torch.Tensor:
    u = uniform01(seed, batch * n_classes)
    return torch.from_numpy((u < p).astype(np.float32)).reshape(batch, n_classes)


def fill_state_dict_(sd: dict, seed: int) -> dict:
    """Deterministically fill a state_dict in place (keys sorted, so layout-independent).

    conv / linear weights: zero-mean, std = sqrt(2 / fan_in)-like (activations stay O(1));
    norm weights in [-0.3, 1.5], biases in [-0.2, 0.2]; running_mean in [-0.5, 0.5],
    running_var in [0.5, 1.5]; key_rel_* get mean dk^-1/2-free unit-scale values.
    """
    for i, k in enumerate(sorted(sd.keys())):
        t = sd[k]
        s = seed * 100003 + i
        if k.endswith("num_batches_tracked"):
            t.zero_()
        elif k.endswith("running_mean"):
            t.copy_(uniform(s, t.shape, -0.5, 0.5))
        elif k.endswith("running_var"):
            t.copy_(uniform(s, t.shape, 0.5, 1.5))
        elif t.dim() == 1 and k.endswith("weight"):
            t.copy_(uniform(s, t.shape, -0.3, 1.5))     # some negative gammas on purpose
        elif t.dim() == 1:
            t.copy_(uniform(s, t.shape, -0.2, 0.2))
        elif "key_rel" in k:
            t.copy_(uniform(s, t.shape, -1.0, 1.0))
        else:
            fan_in = int(np.prod(t.shape[1:]))
            t.copy_(symmetric(s, t.shape, std=(2.0 / fan_in) ** 0.5))
    return sd


def smooth_state_dict_(sd: dict, bias: float, gain_seed: int = 7) -> dict:
    """The well-conditioned ("smooth") regime of the parity fixtures, applied in place to a filled state_dict: every BatchNorm
    gain in [0.8, 1.2], every BatchNorm bias = `bias` (2.5 for the DenseNets, 1.0 for ResNets / EfficientNets: most
    activations stay on the linear side of their non-linearity, so storage rounding does not flip ReLU / max-pool decisions);
    convolution / linear weights keep their kaiming-scale hash fill.  A BatchNorm is recognised by its running_mean entry."""
    for k in list(sd.keys()):
        prefix, _, leaf = k.rpartition(".")
        if prefix + ".running_mean" not in sd:
            continue
        if leaf == "bias":
            sd[k] = torch.full_like(sd[k], float(bias))
        elif leaf == "weight":
            sd[k] = uniform(gain_seed, sd[k].shape, 0.8, 1.2)
    return sd
