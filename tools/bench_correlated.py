#!/usr/bin/env python3
"""BASELINE configs 3 and 5 on one GPU: ms per correlated update and the fraction of photons re-traced.
(bench.py measures config 2; these are the correlated legs of the same path.)"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)

def sync_ms(fn, reps=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r

out = {}
# ---- config 3: TF edit (point 4: 0.2218 -> 0.26), 100 % and 25 % per evaluation
vol = S.heterogeneous_volume(256)
base = list(S.WORKSPACE_TF_POINTS); edit = list(base); edit[3] = (0.26,) + base[3][1:]
for pct, exact in ((100.0, False), (25.0, False), (100.0, True)):
    cm = P.CorrelatedPhotonMapper(ctx, vol, S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0),
                                  tf_points=base, max_incremental_percent=pct, exact_update=exact)
    cm.full_frame(); cm.full_frame()
    full_ms, _ = sync_ms(cm.full_frame, 20)
    res = []
    for rep in range(6):  # alternate edit / revert so that every update re-traces
        pts = edit if rep % 2 == 0 else base
        t_imp, _ = sync_ms(lambda: cm.set_transfer_function(pts))
        t_upd, n = sync_ms(cm.correlated_update)
        rounds = 1
        t_more = 0.0
        while cm.remaining > 0:
            tm, _ = sync_ms(cm.continue_update); t_more += tm; rounds += 1
        res.append((t_imp, t_upd, n / cm.n, rounds, t_more, cm.last_path))
    r = res[2:]
    out[f"config3_{int(pct)}pct" + ("_exact_update" if exact else "")] = {"full_frame_ms": round(full_ms, 4), "importance_ms": round(np.mean([x[0] for x in r]), 4),
                                       "first_update_ms": round(np.mean([x[1] for x in r]), 4), "fraction_first_batch": round(np.mean([x[2] for x in r]), 4),
                                       "rounds": r[0][3], "continuation_ms": round(np.mean([x[4] for x in r]), 4), "light_volume_path": r[0][5]}
    del cm
# ---- config 5: 32-step sequence, blob moving along x
n_steps = 32
vols = [S.heterogeneous_volume(256, S.sequence_blob_center(t, n_steps)) for t in range(n_steps)]
cm = P.CorrelatedPhotonMapper(ctx, vols[0], S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0), tf_points=base)
cm.full_frame()
dvols = [torch.from_numpy(v).to(ctx.device) for v in vols]   # resident: the upload is not part of the update
# the host -> device upload of one time step (16 MiB), reported separately: pinned and pageable host memory
pinned = torch.from_numpy(vols[1]).pin_memory()
staging = torch.empty_like(dvols[1])
up_pinned, _ = sync_ms(lambda: staging.copy_(pinned, non_blocking=True), 10)
up_pageable, _ = sync_ms(lambda: staging.copy_(torch.from_numpy(vols[2])), 10)
steps = []
for t in range(1, n_steps):
    t_vol, _ = sync_ms(lambda: cm.set_volume(dvols[t]))
    t_upd, n = sync_ms(cm.correlated_update)
    steps.append((t_vol, t_upd, n / cm.n, cm.last_path))
out["config5_sequence"] = {"steps": n_steps - 1, "volume_upload_ms_pinned": round(up_pinned, 4), "volume_upload_ms_pageable": round(up_pageable, 4), "volume_step_ms(diff+minmax+importance)": round(np.mean([x[0] for x in steps]), 4),
                           "update_ms": round(np.mean([x[1] for x in steps]), 4), "fraction_retraced": round(np.mean([x[2] for x in steps]), 4),
                           "paths": sorted(set(x[3] for x in steps))}
print(json.dumps(out))
