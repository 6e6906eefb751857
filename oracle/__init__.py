"""CPU oracle for the chexpert conv hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a CPU restatement (plain PyTorch CPU ops, fp32/fp64) of the arithmetic that the
reference executes on its hot path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and there only as the checker.  The product
package `chexpert_amd` never imports it and has no CPU fallback.

Pinning: the restatement is checked against the REAL reference (imported from /root/reference in
the build container by `tests/golden/make_golden.py`, which also writes the fixtures under
`tests/golden/`).  `tests/test_oracle_golden.py` re-checks the oracle against those committed
fixtures on any machine, without the reference present.

Reference files restated here (all paths relative to /root/reference):
  nets.py     models/attn_aug_conv.py:159-304 (Bottleneck/ResNet), :411-517 (_Transition/DenseNet),
              torchvision-0.3.0 _DenseLayer/_DenseBlock semantics (external),
              models/efficientnet.py:27-228
  aaconv.py   models/attn_aug_conv.py:19-100 (AAConv2d)
  step.py     chexpert.py:159-165, :530 (loss + optimiser step), :461-502 (optimiser wiring)
  metrics.py  chexpert.py:130-146 (sklearn roc_curve/auc)
  gradcam.py  chexpert.py:260-303
"""
