import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.argv = [sys.argv[0], "6"]
import torch
exec(open(os.path.join(R, "tools", "wsd_check.py")).read().split("CASES = [")[0])
for name, K, dil, pad, Lin, nb, ep, g2 in [("dec0", 5, 1, 2, 20160, None, 1, True), ("tdnn3", 3, 2, 0, 20152, "bn", 2, False)]:
    nm, run, _ = case(name, K, dil, pad, Lin, nb, ep, g2)
    ops.conv_impl(ws=False); ref = run()
    ops.conv_impl(ws=True); got = run()
    torch.cuda.synchronize()
    r, o = ref[0].float(), got[0].float()
    bad = ((r - o).abs() > 0)
    print(name, "y bad frac", bad.float().mean().item())
    if bad.any():
        rows = bad.any(dim=2)                      # [B, L]
        per_tile = rows.view(B, -1)[:, :rows.shape[1] // 64 * 64].view(B, -1, 64)
        print("  bad rows per (b): ", rows.sum(1).tolist())
        t = per_tile.any(dim=2)
        print("  bad tiles b=0:", t[0].nonzero().flatten().tolist()[:40], "count", int(t[0].sum()))
        bt = t[0].nonzero().flatten().tolist()
        if bt:
            tt = bt[len(bt) // 2]
            print("  tile", tt, "bad rows in tile:", per_tile[0, tt].nonzero().flatten().tolist())
            l0 = tt * 64 + per_tile[0, tt].nonzero().flatten().tolist()[0]
            print("  row", l0, "bad cols", bad[0, l0].nonzero().flatten().tolist()[:40])
            print("  ref", r[0, l0, :4].tolist(), "got", o[0, l0, :4].tolist())
    r, o = ref[1].float(), got[1].float()      # stats [B, nt, 128, 2]
    d = (r - o).abs() / (r.abs().amax() + 1e-30)
    badt = (d > 1e-5).any(dim=3).any(dim=2)
    print(name, "stats bad tiles per b:", badt.sum(1).tolist())
    for b in range(min(B, 2)):
        print("   b", b, badt[b].nonzero().flatten().tolist()[:40])
    if badt.any():
        b, t = badt.nonzero()[0].tolist()
        print("   first bad (b,t)", b, t, "comp0 maxdiff", (r[b, t, :, 0] - o[b, t, :, 0]).abs().max().item(), "comp1 maxdiff", (r[b, t, :, 1] - o[b, t, :, 1]).abs().max().item())
        print("   ref", r[b, t, :3].tolist(), "got", o[b, t, :3].tolist())
