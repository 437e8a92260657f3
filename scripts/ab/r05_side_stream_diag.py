"""Is the step-3 loss gap of test_train_step_fp32 (side stream on / off) a race or Adam amplifying atomics-order noise?  Compare the
step-2 GRADIENTS of the two runs (same parameters to ~1e-7 at that point) tensor by tensor, several trials, allocator warmed up."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from cds_mvsnet_amd import CDSMVSNet, seeded_init_, train as T
from test_train_harness import _train_sample
dev = torch.device("cuda")
sample = _train_sample(dev)
junk = [torch.randn(1 << (10 + i % 12), device=dev) for i in range(200)]      # a used allocator
del junk[::2]
for trial in range(6):
    grads, losses = {}, {}
    for side in (False, True):
        T.SIDE_STREAM_WGRAD = side
        model = seeded_init_(CDSMVSNet(refine=False, ndepths=(48, 32, 8), depth_interals_ratio=(4.0, 2.0, 1.0)), 7).to(dev)
        opt = T.make_optimizer(model, lr=1e-3)
        ls = [T.train_step(model, opt, sample, temperature=0.1, reducer=T.GradAllReducer(model.parameters())) for _ in range(2)]
        grads[side] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        ls.append(T.train_step(model, opt, sample, temperature=0.1, reducer=T.GradAllReducer(model.parameters())))
        losses[side] = [l for l, _ in ls]
    worst = sorted(((float((grads[True][n] - grads[False][n]).abs().max() / (grads[False][n].abs().max() + 1e-30)), n) for n in grads[False]), reverse=True)[:3]
    print(f"trial {trial}: losses off {losses[False]} on {losses[True]}  step-3 rel gap {abs(losses[True][2] - losses[False][2]) / losses[False][2]:.2e}; "
          f"worst step-2 gradient gaps (max abs / max abs): {[(f'{g:.1e}', n) for g, n in worst]}")
