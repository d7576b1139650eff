import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mx_probe import A, B, Af, Bf, pack, run, kmap_contig, ref  # noqa
torch.set_printoptions(precision=2, linewidth=250, sci_mode=False)
for l0 in (0, 1, 16, 17, 35):
    a, b, sa, sb = pack(kmap_contig, 127, 127)
    sa[l0] = 128
    delta = run(a, b, sa, sb) - ref
    rows = (delta.abs().sum(1) > 1e-3).nonzero().flatten().tolist()
    cols = (delta.abs().sum(0) > 1e-3).nonzero().flatten().tolist()
    print(f"A-scale lane {l0}: rows changed {rows}, cols changed {len(cols)}")
    # which k range explains it? least squares against per-(row,16-chunk) candidates
    i = rows[0] if rows else 0
    chunks = torch.stack([Af[i, 8 * c:8 * c + 8] @ Bf[8 * c:8 * c + 8, :] for c in range(16)])  # [16 chunks of 8 k, 16 cols]
    sol = torch.linalg.lstsq(chunks.T, delta[i].unsqueeze(1)).solution.flatten()
    print("   coefficients per 8-wide k chunk:", [round(x, 2) for x in sol.tolist()])
