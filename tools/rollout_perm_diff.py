"""Which module first breaks "a permuted batch gives the permuted result, bit for bit" in the world model's
inference forward?  Forward hooks record a per-clip integer checksum of every module output whose leading dimension
is the batch; the second (permuted) run compares against the first.
usage: rollout_perm_diff.py [B] [horizon] [encoder_depth]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import surgical_world_model as swm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.manual_seed(3)
model = swm.SurgicalActionConditionedWorldModel(img_size=256, num_frames=16, encoder_dim=1024, encoder_depth=depth,
                                                encoder_heads=16, predictor_dim=1024, predictor_depth=12,
                                                predictor_heads=8).eval().cuda()
g = torch.Generator().manual_seed(21)
video = torch.randn(B, 3, 16, 256, 256, generator=g).cuda()
actions, states = torch.randn(B, 16, 7, generator=g).cuda(), torch.randn(B, 16, 7, generator=g).cuda()
perm = torch.randperm(B, generator=g).cuda()

log, mode = [], ["record"]
cursor = [0]
bad = []


def checksum(t):
    if t.dtype == torch.float32:
        return t.contiguous().view(B, -1).view(torch.int32).to(torch.int64).sum(dim=1)
    return t.contiguous().view(B, -1).float().sum(dim=1).view(torch.int32).to(torch.int64)


def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for j, t in enumerate(outs):
            if not (torch.is_tensor(t) and t.dim() >= 2 and t.shape[0] == B and t.is_floating_point()):
                continue
            c = checksum(t)
            if mode[0] == "record":
                log.append((name, j, c))
            else:
                n0, j0, c0 = log[cursor[0]]
                cursor[0] += 1
                assert (n0, j0) == (name, j), (n0, name)
                nbad = int((c0[perm] != c).sum())
                if nbad:
                    bad.append((cursor[0] - 1, name, j, nbad))
    return f


for name, mod in model.named_modules():
    if name:
        mod.register_forward_hook(hook(name))
with torch.no_grad():
    out0 = model(video, actions, states, predict_horizon=horizon)
    mode[0] = "check"
    out1 = model(video[perm].contiguous(), actions[perm].contiguous(), states[perm].contiguous(), predict_horizon=horizon)
print(f"{len(log)} recorded outputs, {len(bad)} differ")
for b in bad[:25]:
    print("  first differing:", b)
same = all(torch.equal(out1[0][h][k], out0[0][h][k][perm]) for h in range(horizon) for k in out0[0][h])
print("predictions bit-identical under permutation:", same, "| logits:", torch.equal(out1[1], out0[1][perm]),
      torch.equal(out1[2], out0[2][perm]))
