"""Per-layer FeatureNet error (GPU layer fed with the oracle's input) — diagnostic, run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cds_mvsnet_amd import FeatureNet, seeded_init_, ops
from cds_mvsnet_amd.model import _FeatureRunner
from oracle import cds_oracle as O
import numpy as np
g = np.load("tests/golden/g5_featurenet.npz")
img = torch.from_numpy(g["img"]); epi_t = torch.from_numpy(g["epipole"])
net = seeded_init_(FeatureNet(8), 7).eval()
sd = {"feature." + k: v.clone() for k, v in net.state_dict().items()}
dev = torch.device("cuda:0")
run = _FeatureRunner(net.to(dev))
p = run.packed.get(run._pack)
epi = (float(epi_t[0,0]), float(epi_t[0,1]))
for T in (1.0, 0.01):
    x = img.unsqueeze(0)
    e = epi_t; ee = epi
    chain = [("conv00",(3,7,11),1),("conv01",(3,5,7),1),("downsample1",None,2),("conv10",(3,5),2),("conv11",(3,5),2),("downsample2",None,4),("conv20",(1,3),4),("conv21",(1,3),4)]
    for name, sizes, div in chain:
        et = e / div; eg = (ee[0]/div, ee[1]/div)
        if sizes is None:
            y_ref = O._plain_block(x, sd, "feature."+name, 2, 1)
            y = run._plain_unit(p, name, x[0].to(dev).contiguous()).cpu()
            print(T, name, "out err", (y - y_ref[0]).abs().max().item())
        else:
            pre_ref, nc_ref = O.dynamic_conv(x, et, T, sd, "feature."+name+".conv", sizes)
            y_ref = F.leaky_relu(F.instance_norm(pre_ref, eps=1e-5), 0.1)
            pre, nc = run._dynamic(p, name, getattr(net, name).conv, x[0].to(dev).contiguous(), eg, T, stats_slope=None)
            y = ops.instnorm_act(pre, ops.ACT_LEAKY01).cpu()
            y2 = ops.instnorm_act(pre_ref[0].to(dev).contiguous(), ops.ACT_LEAKY01).cpu()
            print(T, name, "pre err", (pre.cpu()-pre_ref[0]).abs().max().item(), "nc err", (nc.cpu()-nc_ref[0,0]).abs().max().item(),
                  "out err", (y - y_ref[0]).abs().max().item(), "instnorm-only err", (y2 - y_ref[0]).abs().max().item(), "|pre|max", pre_ref.abs().max().item())
        x = y_ref
