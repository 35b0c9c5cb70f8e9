import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_train as TT
from samplenerfro_amd import _lib
from samplenerfro_amd.train import TrainState, train_step
for Nf, bwd, prec in ((12, "f16", "f16"), (0, "f16", "f16"), (12, "f16", "f16x3"), (12, "bf16", "f16")):
    outs = []
    for whole in (False, True):
        model, state, batch, flags, ev = TT._setup(Nf)
        model.precision = model.eval_precision = _lib.PRECISIONS[prec]
        model._packed = {}
        flags.backward_precision = bwd
        state = TrainState.create(model, state.variables, flags)
        rng = np.array([1, 2], np.uint32)
        taps = None if whole else {}
        train_step(model, rng, state, batch, flags, taps=taps)
        g = state.grads[:state.theta.numel()].clone() if whole else taps["grads"].clone()
        outs.append((g, dict(state.segments)))
    (a, seg), (b, _) = outs
    print(Nf, bwd, prec, {k: float((a[lo:hi] - b[lo:hi]).abs().max()) for k, (lo, hi) in seg.items()}, "max|g|", float(a.abs().max()))
