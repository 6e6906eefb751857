"""Oracle (test infrastructure): attention-augmented convolution, closed form.

Restates /root/reference/models/attn_aug_conv.py:65-97 (AAConv2d.forward) without the
pad/flatten/reshape "relative -> absolute" skewing of :43-63.  With q~ = q * dkh^-1/2:

    S[b,n,(i,j),(k,l)] = sum_d q~[b,n,d,i,j] * ( K[b,n,d,k,l]
                                                 + key_rel_h[d, k-i+H-1]
                                                 + key_rel_w[d, l-j+W-1] )
    P = softmax over (k,l);   O[b, n*dvh+d, i, j] = sum_(k,l) P * V[b,n,d,k,l]

then out_proj (1x1) on O and channel-concat behind the k x k conv branch.
"""
import torch
import torch.nn.functional as F


def _rel_index(L: int) -> torch.Tensor:
    """idx[i, k] = k - i + L - 1  (position of key k relative to query i)."""
    r = torch.arange(L)
    return r[None, :] - r[:, None] + (L - 1)


def attention_logits(q, k, key_rel_h=None, key_rel_w=None):
    """q (already scaled), k: (B, nh, dkh, H, W) -> logits (B, nh, H, W, H, W)."""
    B, nh, dkh, H, W = q.shape
    logits = torch.einsum("bndij,bndkl->bnijkl", q, k)
    if key_rel_h is not None:
        RH = key_rel_h[:, _rel_index(H)]                      # (dkh, H_i, H_k)
        RW = key_rel_w[:, _rel_index(W)]                      # (dkh, W_j, W_l)
        rel_h = torch.einsum("bndij,dik->bnijk", q, RH)       # (B,nh,H,W,H_k)
        rel_w = torch.einsum("bndij,djl->bnijl", q, RW)       # (B,nh,H,W,W_l)
        logits = logits + rel_h[..., :, None] + rel_w[..., None, :]
    return logits


def aaconv2d(x, conv_w, qkv_w, out_w, key_rel_h, key_rel_w, *, stride, dk, dv, nh,
             padding=None, return_weights=False):
    """x (B,C,H,W) -> (B, C_out, H/stride, W/stride); parameter names as the reference state_dict
    (`conv.weight`, `in_proj_qkv.weight`, `out_proj.weight`, `key_rel_h`, `key_rel_w`)."""
    ksz = conv_w.shape[-1] if conv_w is not None else 1
    if padding is None:
        padding = ksz // 2
    qkv = F.conv2d(x, qkv_w, stride=stride)
    B, _, H, W = qkv.shape
    dkh, dvh = dk // nh, dv // nh
    q = qkv[:, :dk].reshape(B, nh, dkh, H, W) * dkh ** -0.5
    k = qkv[:, dk:2 * dk].reshape(B, nh, dkh, H, W)
    v = qkv[:, 2 * dk:].reshape(B, nh, dvh, H, W)
    logits = attention_logits(q, k, key_rel_h, key_rel_w)
    P = torch.softmax(logits.reshape(B, nh, H * W, H * W), dim=-1)
    o = torch.einsum("bnqk,bndk->bndq", P, v.reshape(B, nh, dvh, H * W)).reshape(B, dv, H, W)
    o = F.conv2d(o, out_w)
    out = o if conv_w is None else torch.cat([F.conv2d(x, conv_w, stride=stride, padding=padding), o], 1)
    return (out, P) if return_weights else out


def aa_dims(out_channels: int, k: float, v: float, nh: int):
    """dk/dv rule shared by _Transition / Bottleneck / BasicBlock
    (/root/reference/models/attn_aug_conv.py:418-419, :172-173, :123-124)."""
    dk = max(20 * nh, int((k * out_channels // nh) * nh))
    dv = int((v * out_channels // nh) * nh)
    return dk, dv
