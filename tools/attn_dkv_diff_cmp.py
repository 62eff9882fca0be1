import sys, torch
a, b = torch.load(sys.argv[1]).float(), torch.load(sys.argv[2]).float()
N = int(sys.argv[3]); D = 1024
for name, lo in (("dK", D), ("dV", 2 * D)):
    x, y = a[:, lo:lo + D], b[:, lo:lo + D]
    bad = ~(x == y) | torch.isnan(y)
    print(name, "mismatching elements", int(bad.sum()), "of", bad.numel(), "NaN", int(torch.isnan(y).sum()))
    if bad.any():
        rows = bad.any(dim=1).view(-1, N)            # [B, N]
        print("  bad keys per 32-key group of the first clip:", rows[0].view(-1, 32).sum(dim=1).tolist())
        cols = bad.any(dim=0).view(16, 64)
        print("  bad columns per head (of 64):", cols.sum(dim=1).tolist())
        print("  bad columns inside head 0:", cols[0].nonzero().flatten().tolist())
        i = bad.nonzero()[0].tolist(); print("  first bad element", i, float(x[i[0], i[1]]), float(y[i[0], i[1]]))
