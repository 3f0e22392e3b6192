#!/usr/bin/env python
"""Root cause of round 3's red GPU test (tests/test_gpu_train.py, exact-count trainer vs sync-free trainer, six Adam steps,
gate 5e-3 x max|p| on every parameter entry): run the SAME comparison R times, keep every iteration's gradient / moments /
parameter of both trainers, and for every repetition whose largest parameter difference exceeds the old gate print the history of
the offending entry.  Also measures, per repetition, what one iteration from EQUAL state differs by (the quantity the new tests
gate): max |ga - gb| / max|ga| per group, and the largest step difference of entries with a solid gradient, in units of lr.

    python tools/stress_trainer.py [R=60] > gpurun_out/stress_trainer.log
"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from test_gpu_train import _bg_scene, _group_lr
from gaussianmesh_amd.train import Trainer

R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FREE_ONLY = len(sys.argv) > 2 and sys.argv[2] == "free"      # only part (1), for long hunts
build, bg, cams = _bg_scene()
zero = torch.zeros(3, device="cuda")
dev = torch.device("cuda", torch.cuda.current_device())
old_gate_fail = 0
worst_noise = {}
worst_solid = {}
worst_free = {}
for rep in range(R):
    torch.manual_seed(rep)
    gt = torch.rand((3, 96, 160), device="cuda")
    # ---- (1) the round-3 test: free running
    ta = Trainer(build(), densify_stats=True, sync_free=False, bg_gaussian=bg)
    tb = Trainer(build(), densify_stats=True, sync_free=True, bg_gaussian=bg)
    ta.keep_grads = tb.keep_grads = True
    hist = []
    for i in range(6):
        if i == 4:
            tb.sync_state.capacity[dev] = 64
        ta.step(cams[i % 5], gt, zero); tb.step(cams[i % 5], gt, zero)
        snap = {}
        for t, tag in ((ta, "a"), (tb, "b")):
            for gr in t.optimizer.param_groups:
                snap[(tag, gr["name"])] = (t.last_grads[gr["name"]].clone(), gr["m"][0].clone(), gr["values"][0].clone(), gr["params"][0].detach().clone())
        hist.append(snap)
    for ga, gb in zip(ta.optimizer.param_groups, tb.optimizer.param_groups):
        name = ga["name"]
        p, q = ga["params"][0].detach(), gb["params"][0].detach()
        d = (p - q).abs()
        gate = 5e-3 * max(float(p.abs().max()), 1.0)
        rel = float(d.max()) / _group_lr(ga)
        worst_free[name] = max(worst_free.get(name, 0.0), rel)
        if float(d.max()) > gate:
            old_gate_fail += 1
            flat = int(d.reshape(-1).argmax())
            print("rep %d: group %s shape %s: max |p-q| = %.4g > old gate %.4g (= %.2f lr); entry %d history (exact | sync-free):"
                  % (rep, name, tuple(p.shape), float(d.max()), gate, rel, flat))
            for i, snap in enumerate(hist):
                A, B = snap[("a", name)], snap[("b", name)]
                gmax = float(A[0].abs().max())
                print("   it %d  grad % .3e | % .3e  (tensor max %.3e, entry/max %.1e)   m % .3e | % .3e   v %.3e | %.3e   p % .6f | % .6f"
                      % (i, float(A[0].reshape(-1)[flat]), float(B[0].reshape(-1)[flat]), gmax, abs(float(A[0].reshape(-1)[flat])) / gmax,
                         float(A[1].reshape(-1)[flat]), float(B[1].reshape(-1)[flat]), float(A[2].reshape(-1)[flat]), float(B[2].reshape(-1)[flat]),
                         float(A[3].reshape(-1)[flat]), float(B[3].reshape(-1)[flat])))
    if FREE_ONLY:
        if rep % 100 == 99:
            print("rep", rep, "done", flush=True)
        continue
    # ---- (2) one iteration from equal state, every iteration
    ta = Trainer(build(), densify_stats=True, sync_free=False, bg_gaussian=bg)
    tb = Trainer(build(), densify_stats=True, sync_free=True, bg_gaussian=bg)
    ta.keep_grads = tb.keep_grads = True
    for i in range(6):
        tb.copy_state_from(ta)
        before = {gr["name"]: gr["params"][0].detach().clone() for gr in ta.optimizer.param_groups}
        la, pa = ta.step(cams[i % 5], gt, zero); lb, pb = tb.step(cams[i % 5], gt, zero)
        assert torch.equal(pa["render"], pb["render"]) and torch.equal(la, lb), (rep, i)
        for ga, gb in zip(ta.optimizer.param_groups, tb.optimizer.param_groups):
            name = ga["name"]
            g_a, g_b = ta.last_grads[name], tb.last_grads[name]
            gmax = float(g_a.abs().max())
            worst_noise[name] = max(worst_noise.get(name, 0.0), float((g_a - g_b).abs().max()) / gmax)
            solid = (g_a.abs() >= 1e-3 * gmax) & (g_b.abs() >= 1e-3 * gmax)
            da, db = ga["params"][0].detach() - before[name], gb["params"][0].detach() - before[name]
            worst_solid[name] = max(worst_solid.get(name, 0.0), float((da - db)[solid].abs().max()) / _group_lr(ga))
    if rep % 10 == 9:
        print("rep", rep, "done", flush=True)
print("repetitions", R, "| parameter tensors over the round-3 gate (5e-3 x max|p|) in the free-running comparison:", old_gate_fail)
print("free running, worst |p-q| after six steps, in units of the group's lr:", {k: round(v, 3) for k, v in worst_free.items()})
print("equal state, worst |ga-gb| / max|ga| per group:", {k: float("%.2e" % v) for k, v in worst_noise.items()})
print("equal state, worst step difference of solid-gradient entries, in lr:", {k: float("%.2e" % v) for k, v in worst_solid.items()})
