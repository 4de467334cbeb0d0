#!/usr/bin/env python3
"""bench.py -- LM iterations/sec of the MI355X photometric BA engine on the BASELINE.json workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one trust-region LM iteration of the reference's solve (reference src/photobundle.cc:829): damped Schur
solve of the stored linearisation + candidate cost pass, and -- when the step is accepted -- the Jacobian pass at the
new point.  Workload at N = 1: BASELINE.json configs[1] (8-frame window, 50k points, 5x5 patch, single level, synthetic
KITTI-shaped frames).  N > 1 is WEAK scaling: every rank owns 50k points of ONE N*50k-point window, all
cameras/frames replicated, the reduced camera system and the step scalars exchanged once per step; `value` is the
LM iterations per second OF THAT WINDOW (it is NOT multiplied by N: an exchange that doubled the step time would halve
it), `residuals_per_sec` -- scalar residual evaluations of all ranks per second -- is the quantity that grows with N.
With N > 1 the same launch also appends a `strong` record: BASELINE.json configs[3] (ONE 16-frame x 200k-point window)
solved by rank 0 alone and then point-sharded over the N ranks, time per LM iteration of each and their ratio
(PBA_BENCH_STRONG=0 skips it).  Inputs are resident in HBM before the timed region.

--config 2 is configs[2], the 3-level pyramid path (reference src/photobundle_pyramid.cc:9-69): one window per level (1241x376,
621x188, 311x94; calibration halved per level), every level timed like the headline, plus the device-side pyramid build; one JSON
line whose `levels` list carries the per-level iteration time, kernel shares and roofline figures.

--config 3 is configs[3] as the STRONG-scaling workload itself (`value` = LM iterations per second of the one window);
`--config 3 --emulate-rank-of R` runs, on ONE GPU, the full window AND one rank's 1/R shard through the multi-rank code
path (peer exchange at world = 1) and prints the projected R-GPU speed-up; --config 4 is configs[4] (11x11 patches +
Huber 0.05, weak scaling like configs[1]).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # bytes/s, MI355X HBM3E spec (MI355X_MICROARCH.md)


def algorithmic_bytes(radius, n_bar, channels=1):
    """SURVEY.md 8d byte model per observation (C channels)."""
    F = (2 * radius + 2) ** 2 * channels
    P = (2 * radius + 1) ** 2 * channels
    sample_jac = 12 * F + 4 * P + 24 + 8          # footprint (I, Gx, Gy fp32) + descriptor + XYZ + obs index
    schur = 144 + 72.0 / n_bar                    # W_pc write + V_p, g_p per point
    b_jac = sample_jac + schur
    b_cost = 4 * F + 4 * P + 24 + 8
    b_res = 144 + 72.0 / n_bar
    return dict(sample_jac=sample_jac, schur=schur, b_jac=b_jac, b_cost=b_cost, b_res=b_res)


def kernel_source_id():
    """sha1 (12 hex digits) over the device sources: ties profiles/traffic.json (tools/collect_profiles.py stores the id of the
    build its counters were taken on) to the build that prints the line."""
    h = hashlib.sha1()
    for f in ("pba_kernels.h", "pba_solve.h", "pba_resident.h", "pba_device.h", "pba_engine.hip"):
        with open(os.path.join(ROOT, "photobundle_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def usable_cores():
    """Cores this process may actually run on: the affinity mask capped by the cgroup CPU quota (the GPU boxes of the pool show 256
    hardware threads and a quota of 16: OpenMP teams beyond the quota only time-slice)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def bench_pyramid(args):
    """BASELINE.json configs[2]: 3-level pyramid (photobundle_pyramid path), 8-frame window, 50k points, 5x5 patch, 1 GPU.
    The reference runs one PhotometricBundleAdjustment per level, coarse to fine, every level selecting its own points on its own
    image (src/photobundle_pyramid.cc:34-66); here every level gets a synthetic window of its own size with the calibration halved per
    level (Calibration::pyrDown) and as many points as the level holds, up to 50k (a 311x94 image has 29k pixels).  One "step" = one LM
    iteration on EACH level; `value` = such coarse-to-fine iterations per second, `levels` carries every level on its own."""
    import torch
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--config 2 is a single-GPU workload (the pyramid levels of one window run one after the other)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(0)
    from photobundle_amd import synthetic
    from photobundle_amd.engine import Engine, default_solver_options
    frames = args.frames or 8
    radius = args.radius or 2
    want = args.points or 50000
    size, K = synthetic.KITTI_SIZE, synthetic.KITTI_K
    t0 = time.time()
    levels = []
    for lvl in range(3):
        rows, cols = size
        Kl = tuple(v * 0.5 ** lvl for v in K)
        for _ in range(lvl):
            rows, cols = (rows + 1) // 2, (cols + 1) // 2
        # points the level can hold: sites are distinct pixels that stay inside the image in all frames (synthetic.make_window)
        n_pts = min(want, int(0.40 * (rows - 2 * (radius + 2)) * (cols - 2 * (radius + 2))))
        prob = synthetic.make_window(n_frames=frames, n_points=n_pts, radius=radius, size=(rows, cols), K=Kl, huber=args.huber or 0.0)
        levels.append(dict(level=lvl, rows=rows, cols=cols, prob=prob))
    t_gen = time.time() - t0

    def opts(k):
        return default_solver_options(max_num_iterations=k, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)

    raw_buffers = Engine.solve_buffers()
    out_levels, total_s, total_bytes = [], 0.0, 0.0
    engines = []
    for L in levels:
        prob = L["prob"]
        eng = Engine(L["rows"], L["cols"], prob.K, prob.radius, prob.n_frames, huber=prob.huber, device=0)
        eng.load(prob)
        engines.append(eng)

        def reset(eng=eng, prob=prob):
            eng.set_problem(prob.xyz, prob.desc, prob.obs_point, prob.obs_slot, prob.weights)
            eng.set_cameras(prob.cams, prob.fixed_slot)
        if args.warmup > 0:
            eng.solve(opts(args.warmup))
        times, res = [], None
        for _ in range(max(1, args.repeats)):
            reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            raw = eng.solve_raw(opts(args.steps), buffers=raw_buffers)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t1)
            res = Engine.unpack_solve(*raw)
        times.sort()
        el = times[len(times) // 2]
        iters = len(res["iterations"]) - 1
        n_succ = sum(1 for i in res["iterations"][1:] if i["step_is_successful"])
        reset()
        eng.set_profiling(2)
        eng.solve(opts(args.steps))
        ctr = eng.counters()
        n_bar = prob.n_obs / prob.n_points
        ab = algorithmic_bytes(prob.radius, n_bar)
        n_jac, n_cost, n_res = res["num_jacobian_passes"], res["num_cost_passes"], res["num_resolve_passes"]
        run_bytes = prob.n_obs * (n_jac * ab["b_jac"] + n_cost * ab["b_cost"] + n_res * ab["b_res"])
        run_bytes_nominal = prob.n_obs * ((1 + n_succ) * ab["b_jac"] + iters * ab["b_cost"] + (iters - n_succ) * ab["b_res"])
        per = lambda k: (1e3 * ctr[k + "_ms"] / ctr[k + "_launches"]) if ctr[k + "_launches"] else 0.0
        kern = {"k_sample<JAC> (Jacobian pass)": (per("linearize"), ab["sample_jac"]), "k_schur (point elimination)": (per("schur"), ab["schur"]),
                "k_reduce_solve (partials + reduced solve)": (per("solve"), 0.0)}
        frame_mb = frames * L["rows"] * L["cols"] * 4 / 1e6
        out_levels.append({
            "level": L["level"], "image": "%dx%d u8" % (L["cols"], L["rows"]), "points": prob.n_points, "observations": prob.n_obs,
            "us_per_iteration": 1e6 * el / max(1, iters), "us_per_iteration_min": 1e6 * times[0] / max(1, iters),
            "us_per_iteration_max": 1e6 * times[-1] / max(1, iters), "iterations": iters, "successful": n_succ,
            "solve_driver": eng.solve_driver(), "initial_cost": res["initial_cost"], "final_cost": res["final_cost"],
            "packed_frames_MB": frame_mb,
            "frames_resident_in": "one XCD's 4 MiB L2" if frame_mb / 8 <= 4.0 else ("the 256 MB Infinity Cache" if frame_mb <= 256 else "HBM"),
            "kernels_us_per_launch": {k: v[0] for k, v in kern.items()},
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
                         "whole_iteration_frac": (run_bytes / el) / HBM_PEAK, "whole_iteration_frac_nominal": (run_bytes_nominal / el) / HBM_PEAK,
                         "algorithmic_bytes_per_obs": {"b_jac": ab["b_jac"], "b_cost": ab["b_cost"], "b_res": ab["b_res"]},
                         "per_kernel": {k: {"avg_launch_us": v[0], "algorithmic_bytes_per_obs": v[1],
                                            "achieved": (prob.n_obs * v[1] / (1e-6 * v[0]) / 1e9) if v[0] > 0 else 0.0,
                                            "frac": (prob.n_obs * v[1] / (1e-6 * v[0]) / HBM_PEAK) if v[0] > 0 else 0.0} for k, v in kern.items()},
                         "traffic": None,
                         "traffic_note": "counter passes of this workload: profiles/r06/config2_* (tools/profile_config2.sh)"}})
        total_s += el / max(1, iters)
        total_bytes += run_bytes / max(1, iters)
    # ---- pyramid build: level 0 frames are resident; levels 1, 2 are produced on the device, level to level (pba_set_frame_pyr_down) ----
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    build_ms = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s_ in range(frames):
            engines[1].set_frame_pyr_down(s_, engines[0], s_, want_image=False)
            engines[2].set_frame_pyr_down(s_, engines[1], s_, want_image=False)
        torch.cuda.synchronize()
        build_ms.append(1e3 * (time.perf_counter() - t1))
    build_ms.sort()
    dom = max(out_levels, key=lambda l: l["us_per_iteration"])
    dk = max((k for k in dom["roofline"]["per_kernel"] if dom["roofline"]["per_kernel"][k]["algorithmic_bytes_per_obs"] > 0),
             key=lambda k: dom["roofline"]["per_kernel"][k]["avg_launch_us"])
    out = {
        "metric": "LM iters/sec + residuals/sec, 8-frame KITTI window, 50k pts, 5x5 patch",
        "value": 1.0 / total_s, "unit": "coarse-to-fine LM iterations/s (one iteration on each of the 3 pyramid levels)",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * total_s, "repeats": max(1, args.repeats),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: 3-level pyramid, %d-frame window, up to %d points per level, %dx%d patch, dense visibility"
                               % (frames, want, 2 * radius + 1, 2 * radius + 1),
                   "levels": [l["image"] for l in out_levels], "observations": [l["observations"] for l in out_levels],
                   "sampler_precision": "exact", "channels": 1, "parallelism": "single GPU, the levels one after the other (coarse to fine in the class)"},
        "levels": out_levels,
        "pyramid_build": {"ms_per_window": build_ms[len(build_ms) // 2], "ms_per_frame": build_ms[len(build_ms) // 2] / frames,
                          "what": "cv::pyrDown of %d frames into levels 1 and 2 on the device (k_pyr_down + k_pack_frame, pba_set_frame_pyr_down), host wall clock "
                                  "around the %d calls + synchronize; the class pays it once per NEW frame, i.e. 1/%d of this per addFrame" % (frames, 2 * frames, frames)},
        "roofline": {"bound": "hbm", "kernel": "%s at level %d (%s)" % (dk, dom["level"], dom["image"]),
                     "achieved": dom["roofline"]["per_kernel"][dk]["achieved"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": dom["roofline"]["per_kernel"][dk]["frac"], "traffic": None,
                     "whole_pass_frac": (total_bytes / total_s) / HBM_PEAK, "kernel_source_id": kernel_source_id(), "traffic_matches_build": None},
        "gen_seconds": t_gen,
    }
    print(json.dumps(out))
    for e_ in engines:
        e_.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3, 4),
                    help="BASELINE.json configs[k]: 1 = 8 frames x 50k points x 5x5 per GPU (weak scaling, the headline); "
                         "2 = the 3-level pyramid path: the same window shape at 1241x376, 621x188 and 311x94 (single GPU); "
                         "3 = ONE 16-frame x 200k-point window point-sharded over the N ranks (STRONG scaling: 200k / N points per "
                         "rank, value is not multiplied by N); 4 = 8 frames x 50k points x 11x11 + Huber 0.05 per GPU")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--points", type=int, default=None, help="points per GPU (config 3: points of the whole window)")
    ap.add_argument("--radius", type=int, default=None)
    ap.add_argument("--huber", type=float, default=None)
    ap.add_argument("--visibility", choices=("dense", "causal"), default="dense")
    ap.add_argument("--precision", choices=("exact", "fp32", "bf16"), default="exact",
                    help="sampler precision (configs[4] tolerance sweep); only \"exact\" has reference parity")
    ap.add_argument("--channels", type=int, default=1, choices=(1, 3, 8),
                    help="descriptor channels of a residual block (reference Options::descriptorType): 1 Intensity (the headline), "
                         "3 IntensityAndGradient, 8 BitPlanes")
    ap.add_argument("--inverse-depth", action="store_true",
                    help="the north star's SE(3) x inverse-depth parameterisation (pba_set_inverse_depth; no reference counterpart)")
    ap.add_argument("--repeats", type=int, default=25,
                    help="the K-step solve is timed this many times (each bracketed by barrier + synchronize, state reset outside "
                         "the bracket); `value` / `ms_per_step` come from the MEDIAN repeat, min / max are reported beside it")
    ap.add_argument("--emulate-rank-of", type=int, default=0,
                    help="with --config 3 on ONE GPU: time the full window (single-rank path), then one rank's 1/R shard through the "
                         "multi-rank code path (PBA_FORCE_MULTI: reduction -> mailbox -> solve with the peer wait, k_decide with the peer "
                         "wait) and print `strong_projection` (t_full / (t_rank + assumed xGMI latency of the two exchanges))")
    ap.add_argument("--xgmi-exchange-us", type=float, default=5.0,
                    help="--emulate-rank-of: latency ASSUMED per cross-device exchange on top of the measured single-device path "
                         "(flag write -> remote poll -> R mailbox reads of <= 35 KB over xGMI); no multi-GPU box was available to measure it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the second CPU-baseline leg (all host cores)")
    ap.add_argument("--cpu-points", type=int, default=50000, help="points of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=20)
    args = ap.parse_args()
    if args.config == 2:
        return bench_pyramid(args)
    preset = {1: dict(frames=8, points=50000, radius=2, huber=0.0), 3: dict(frames=16, points=200000, radius=2, huber=0.0),
              4: dict(frames=8, points=50000, radius=5, huber=0.05)}[args.config]
    explicit = any(getattr(args, k) is not None for k in preset)
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    strong = args.config == 3
    emulate = args.emulate_rank_of if (strong and args.gpus == 1) else 0
    if emulate:
        args.no_cpu_baseline = True

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # PBA_BENCH_BACKEND=gloo is a test hook: N ranks sharing the GPUs that exist (RCCL refuses duplicate devices, so
    # this exercises the host-staged fallback and the rest of the multi-rank logic on a 1-GPU box)
    backend = os.environ.get("PBA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    ctl = "cuda" if backend == "nccl" else "cpu"       # device of the control-plane tensors
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from photobundle_amd import synthetic
    from photobundle_amd.engine import Engine, default_solver_options

    # ---- synthetic window: identical frames/cameras on every rank, rank-specific points ----------------------
    t0 = time.time()
    default_shape = not explicit and (args.visibility, args.precision, args.channels, args.inverse_depth) == ("dense", "exact", 1, False)
    ch_fn = synthetic.channel_fn({3: "IntensityAndGradient", 8: "BitPlanes"}[args.channels]) if args.channels > 1 else None
    if strong:
        # the SAME window on every rank (same seeds), each rank keeps its contiguous shard of whole points
        whole = synthetic.make_window(n_frames=args.frames, n_points=args.points, radius=args.radius, huber=args.huber,
                                      visibility=args.visibility, dense_births=(0, 8) if args.frames > 8 else (0,), channel_fn=ch_fn)
        prob = whole.shard(rank, world) if world > 1 else whole
        del whole
    else:
        prob = synthetic.make_window(n_frames=args.frames, n_points=args.points, radius=args.radius, huber=args.huber,
                                     visibility=args.visibility, point_seed_offset=rank, channel_fn=ch_fn)
    t_gen = time.time() - t0
    rows, cols = prob.images.shape[1:]
    P = prob.patch_len * args.channels          # residuals of one block
    n_obs_local = prob.n_obs
    n_bar = n_obs_local / prob.n_points

    rays = rho0 = None
    if args.inverse_depth:
        rays, rho0 = synthetic.inverse_depth_rays(prob)

    def opts(k):
        # fixed iteration count: tolerances disabled (SURVEY.md 8d "Fixed 10 LM iterations for throughput")
        return default_solver_options(max_num_iterations=k, function_tolerance=0.0, gradient_tolerance=0.0,
                                      parameter_tolerance=0.0)

    def new_engine():
        e_ = Engine(rows, cols, prob.K, prob.radius, prob.n_frames, huber=prob.huber, device=local_rank, precision=args.precision,
                    channels=args.channels)
        e_.load(prob)
        if args.inverse_depth:
            e_.set_inverse_depth(rays, rho0)
        return e_

    def all_ranks_ok(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def _allreduce(a, op):          # host-staged transport through torch.distributed (callback of pba_comm_init_callback)
        t = torch.from_numpy(a.copy()).to(ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        a[:] = t.cpu().numpy()

    def join_transport(e_, want_rccl):
        """Collective.  RCCL (device-direct all-reduces on the engine's stream) when every rank can, else host-staged."""
        if want_rccl:
            uid = torch.zeros(129, dtype=torch.uint8, device=ctl)     # 128-byte ncclUniqueId + "valid" byte
            if rank == 0:
                try:
                    raw = bytearray(Engine.comm_unique_id()) + bytearray([1])
                    uid.copy_(torch.frombuffer(raw, dtype=torch.uint8))
                except Exception as exc:
                    print("rank 0: no RCCL unique id (%s)" % exc, file=sys.stderr)
            dist.broadcast(uid, 0)
            uid_host = uid.cpu().numpy()
            ok = bool(uid_host[128])
            if ok:
                try:
                    e_.comm_init_rccl(bytes(uid_host[:128].tobytes()), rank, world)
                except Exception as exc:
                    print("rank %d: RCCL transport unavailable (%s)" % (rank, exc), file=sys.stderr)
                    ok = False
            if all_ranks_ok(ok):
                return e_, "RCCL all-reduce of the reduced camera system"
            if ok:               # mixed outcome: a fresh engine, so that every rank uses the same transport
                e_.close()
                e_ = new_engine()
        e_.comm_init_callback(_allreduce, rank, world)
        return e_, "host-staged all-reduce via torch.distributed" + (" (RCCL init failed)" if want_rccl else "")

    if world > 1:
        os.environ.setdefault("PBA_WAIT_TIMEOUT_S", "30")      # read at pba_create: a stuck exchange is an error after 30 s, not a hang (room for RCCL's lazy first-collective set-up, below the 60 s compute-queue watchdog)
    eng = new_engine()
    transport = "RCCL all-reduce of the reduced camera system"
    if world > 1:
        eng, transport = join_transport(eng, backend == "nccl")
        if os.environ.get("PBA_PEER", "1") != "0":
            # per-step exchanges as device-side mailbox reads over peer-mapped memory (no collective launch per LM step); the
            # call is collective and every rank ends up with the same answer -- the base transport stays when IPC is unavailable.
            # The path is then exercised once before anything is timed: should it fail on ANY rank (it has only ever run
            # between two processes on one device), every rank rebuilds its engine on the base transport alone.
            eng.comm_enable_peer_exchange()
            if eng.comm_transport().endswith("+peer"):
                try:
                    eng.solve(opts(2))
                    ok = True
                except Exception as exc:
                    print("rank %d: peer exchange failed its self-test (%s)" % (rank, exc), file=sys.stderr)
                    ok = False
                if not all_ranks_ok(ok):
                    try:
                        eng.close()
                    except Exception:
                        pass
                    eng = new_engine()
                    eng, transport = join_transport(eng, backend == "nccl")
        transport = "%s (%s)" % (transport, eng.comm_transport())
    elif os.environ.get("PBA_FORCE_MULTI") == "1":
        # diagnostics: the multi-rank code path (RCCL all-reduces on the engine's stream, k_decide) at world = 1
        eng.comm_init_rccl(Engine.comm_unique_id(), 0, 1)

    def reset_state():
        eng.set_problem(prob.xyz, prob.desc, prob.obs_point, prob.obs_slot, prob.weights)
        eng.set_cameras(prob.cams, prob.fixed_slot)
        if args.inverse_depth:
            eng.set_inverse_depth(rays, rho0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:      # (one rank: a second synchronize behind the first is 10-20 us of host API time inside the bracket)
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        eng.solve(opts(args.warmup))
        reset_state()
    # EXACTLY args.steps LM iterations are timed.  A solve of this window stops making progress after ~150 iterations
    # (cost change exactly 0), so longer requests are served as consecutive solves of <= CHUNK iterations from the same
    # initial window; the state reset between them (a host upload) is outside the timed region, every timed solve is
    # bracketed by barrier + synchronize.
    CHUNK = 50
    raw_buffers = Engine.solve_buffers()                                   # result structs allocated outside the timed region

    def timed_run():
        """EXACTLY args.steps LM iterations, timed; returns (seconds, pass / iteration counts, last result)."""
        elapsed, remaining = 0.0, args.steps
        tot = dict(iters=0, n_jac=0, n_cost=0, n_res=0, n_succ=0)
        res = None
        while remaining > 0:
            reset_state()
            o_k = opts(min(remaining, CHUNK))
            barrier()
            t1 = time.perf_counter()
            raw = eng.solve_raw(o_k, buffers=raw_buffers)                  # pba_solve; the refined state stays in HBM
            barrier()
            elapsed += time.perf_counter() - t1
            res = Engine.unpack_solve(*raw)                                # C structs -> dicts: bookkeeping, not a step
            done = len(res["iterations"]) - 1
            if done <= 0:
                break
            tot["iters"] += done
            tot["n_jac"] += res["num_jacobian_passes"]; tot["n_cost"] += res["num_cost_passes"]; tot["n_res"] += res["num_resolve_passes"]
            tot["n_succ"] += sum(1 for i in res["iterations"][1:] if i["step_is_successful"])
            remaining -= done
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, tot, res

    # One K-step solve is a few milliseconds: it is repeated (same initial window every time, so every repeat runs the
    # same iterations) and the MEDIAN repeat is the reported one; the whole timed region is repeats x K steps.
    runs = [timed_run() for _ in range(max(1, args.repeats))]
    times = sorted(r[0] for r in runs)
    elapsed = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    tot, res = runs[0][1], runs[0][2]
    # PCIe-inclusive variant (never `value`): the C-ABI receives HOST buffers, so a cold window also pays the upload of
    # all frames (u8), the problem (points, descriptors, observation lists) and the cameras before the same solve
    barrier()
    t2 = time.perf_counter()
    for s_ in range(prob.n_frames):
        if args.channels > 1:
            eng.set_frame_channels(s_, prob.channel_images[s_])
        else:
            eng.set_frame(s_, prob.images[s_])
    reset_state()
    upload_s = time.perf_counter() - t2
    # measured device-copy bandwidth (SURVEY.md 8d asks for it next to the 8 TB/s spec figure): 1 GiB device-to-device
    copy_gbps = None
    if rank == 0:
        try:
            src_t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            dst_t = torch.empty_like(src_t)
            dst_t.copy_(src_t)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                dst_t.copy_(src_t)
            ev1.record()
            torch.cuda.synchronize()
            copy_gbps = 5 * 2.0 * (1 << 30) / (ev0.elapsed_time(ev1) * 1e-3) / 1e9     # read + write
            del src_t, dst_t
        except Exception:
            copy_gbps = None
    # per-kernel shares of the iteration: one more solve, outside the timed region.  Single rank: device time stamps inside
    # the same asynchronous pipeline the timed region ran (interval between consecutive kernel ends; nothing is added to
    # the stream -- an event record after every kernel makes it end with a cache write-back the pipelined run never pays,
    # which inflated k_schur by ~6 us).  More ranks: HIP events around every kernel of the host-stepped driver, which also
    # times the two per-step exchanges.
    reset_state()
    if world == 1:
        eng.set_profiling(2)
        eng.solve(opts(args.steps))
        timing_source = "device time stamps in the asynchronous pipeline (interval between consecutive kernel ends)"
        if eng.solve_driver() == "resident":
            timing_source = ("phase stamps of the serial workgroup of the resident solve (one launch per solve): elimination | reduction + reduced "
                             "solve | back-substitution + sampling + decision, reported under the names of the kernels that do this work on the pipelined path")
    else:
        eng.reset_counters()
        eng.solve(opts(min(args.steps, 10)))
        timing_source = "HIP events around every kernel of the host-stepped driver (each bracket ends with a cache write-back)"
    ctr = eng.counters()

    iters_done = tot["iters"]
    n_jac, n_cost, n_res = tot["n_jac"], tot["n_cost"], tot["n_res"]
    n_obs_global = res["num_residual_blocks"]
    iters_per_sec = iters_done / elapsed
    value = iters_per_sec          # LM iterations per second of the ONE window the ranks solve together (never x N)
    residuals_per_sec = n_obs_global * P * (n_jac + n_cost) / elapsed   # residuals actually evaluated by the engine

    ab = algorithmic_bytes(prob.radius, n_bar, args.channels)
    # SURVEY.md 8d accounting rule, "with the actual pass counts of the run": the engine fuses the candidate cost pass into
    # a speculative Jacobian pass, so a successful iteration is ONE Jacobian pass (and no cost pass); a rejected one is
    # followed by a re-solve from the stored linearisation.
    n_succ = tot["n_succ"]
    n_rej = iters_done - n_succ
    run_bytes = n_obs_global * (n_jac * ab["b_jac"] + n_cost * ab["b_cost"] + n_res * ab["b_res"])
    # ... and with the NOMINAL pass counts of the reference's algorithm for the same trace (iteration 0 = one Jacobian
    # pass, a successful iteration = cost pass + Jacobian pass, a rejected one = cost pass + re-solve): reported beside it
    run_bytes_nominal = n_obs_global * ((1 + n_succ) * ab["b_jac"] + iters_done * ab["b_cost"] + n_rej * ab["b_res"])
    # dominant kernel, timed live with HIP events on the engine's stream (rank-local launch = local observations)
    kern = {
        "k_sample<JAC> (Jacobian pass)": (ctr["linearize_ms"], ctr["linearize_launches"], ab["sample_jac"]),
        "k_sample<cost> (cost pass)": (ctr["cost_ms"], ctr["cost_launches"], ab["b_cost"]),
        "k_schur (point elimination)": (ctr["schur_ms"], ctr["schur_launches"], ab["schur"]),
        # serial stretch of the iteration: reduction of the Schur partials + reduced camera solve (O(frames^2): no per-observation bytes)
        "k_reduce_solve (partials + reduced solve)": (ctr["solve_ms"], ctr["solve_launches"], 0.0),
    }
    # the kernel with the largest average launch duration (no tie-break; `per_kernel` carries the others)
    avg_ms = {k: v[0] / max(1, v[1]) for k, v in kern.items()}
    dom = max((k for k in kern if kern[k][2] > 0), key=lambda k: avg_ms[k])      # (the serial solve stage moves no per-observation bytes)
    ms, launches, bytes_per_obs = kern[dom]
    avg_s = (ms / max(1, launches)) * 1e-3
    achieved = n_obs_local * bytes_per_obs / avg_s if avg_s > 0 else 0.0
    # counter-based figures of the same kernel from the committed PMC passes (profiles/traffic.json, tools/collect_profiles.py):
    # HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, calibration in profiles/r02/fetch_calibration.txt) and the
    # VALU instruction count, which gives the issue floor: wave instructions x 4 cycles / (1024 SIMDs x 2.4 GHz)
    traffic = valu_insts = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic_sha = traffic_src_id = None
    src_id = kernel_source_id()
    if os.path.exists(tpath) and default_shape and args.config == 1:
        try:
            raw = open(tpath, "rb").read()
            traffic_sha = hashlib.sha1(raw).hexdigest()[:12]
            tj = json.loads(raw)
            traffic = tj.get(dom.split(" ")[0])
            valu_insts = tj.get(dom.split(" ")[0] + "_valu_insts")
            traffic_src_id = tj.get("kernel_source_id")
        except Exception:
            traffic = valu_insts = None
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": achieved / HBM_PEAK, "traffic": traffic,
        "traffic_source": ("profiles/traffic.json sha1 %s (committed rocprofv3 --pmc passes of this workload; NOT measured in this run); "
                           "counters taken on kernel sources %s, this build is %s: %s"
                           % (traffic_sha, traffic_src_id, src_id, "MATCH" if traffic_src_id == src_id else "STALE counter file")
                           if traffic else None),
        "traffic_matches_build": (traffic_src_id == src_id) if traffic else None,
        "kernel_source_id": src_id,
        "traffic_frac": (traffic / avg_s / HBM_PEAK) if (traffic and avg_s > 0) else None,
        "valu_floor_us": (valu_insts * 4.0 / (1024 * 2.4e9) * 1e6) if valu_insts else None,
        "avg_launch_us": avg_s * 1e6, "timing_source": timing_source, "algorithmic_bytes_per_obs": bytes_per_obs,
        "whole_iteration_frac": (run_bytes / elapsed) / (HBM_PEAK * world),
        "whole_iteration_frac_nominal": (run_bytes_nominal / elapsed) / (HBM_PEAK * world),
        "kernels_ms_per_launch": {k: (v[0] / max(1, v[1])) for k, v in kern.items()},
        # the same figures for every timed kernel (the two big ones are within a few microseconds of each other)
        "per_kernel": {k: {"avg_launch_us": 1e3 * v[0] / v[1], "algorithmic_bytes_per_obs": v[2],
                           "frac": n_obs_local * v[2] / (1e-3 * v[0] / v[1]) / HBM_PEAK}
                       for k, v in kern.items() if v[1] > 0 and v[0] > 0},
    }

    out = {
        "metric": "LM iters/sec + residuals/sec, 8-frame KITTI window, 50k pts, 5x5 patch",
        "value": value, "unit": ("LM iters/s of the one %d-point window (strong scaling)" % args.points) if strong
                                 else "LM iters/s of the one %d-point window (%dk points per GPU: weak scaling; not multiplied by N)"
                                      % (args.points * world, args.points // 1000),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, iters_done),
        "repeats": len(times), "ms_per_step_min": 1e3 * times[0] / max(1, iters_done),
        "ms_per_step_max": 1e3 * times[-1] / max(1, iters_done), "timed_region_ms": 1e3 * sum(times),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s%d-frame window, %d points%s, %dx%d patch, single level, %s visibility"
                               % (("configs[%d]: " % args.config) if default_shape else "", prob.n_frames, args.points,
                                  " sharded over the ranks" if strong else "/GPU", 2 * prob.radius + 1,
                                  2 * prob.radius + 1, args.visibility),
                   "image": "%dx%d u8" % (cols, rows), "observations": int(n_obs_global), "huber": prob.huber,
                   "sampler_precision": args.precision, "channels": args.channels,
                   "point_parameterisation": "inverse depth on fixed rays (no reference counterpart)" if args.inverse_depth else "free world points (reference)",
                   "parallelism": "points sharded x%d, cameras+frames replicated, %s" % (world, transport)},
        "iters_per_sec": iters_per_sec, "residuals_per_sec": residuals_per_sec,
        "solve_driver": eng.solve_driver(),
        "lm": {"iterations": iters_done, "successful": n_succ, "solves": -(-args.steps // CHUNK), "jacobian_passes": n_jac,
               "cost_passes": n_cost, "resolve_passes": n_res, "initial_cost": res["initial_cost"],
               "final_cost": res["final_cost"], "message": res["message"]},
        "roofline": roofline,
        "exchange": {"transport": eng.comm_transport(), "ranks_seen_by_transport": eng.comm_rank_count(),
                     "us_per_step": (1e3 * ctr["exchange_ms"] / ctr["exchange_launches"]) if ctr["exchange_launches"] else 0.0,
                     "note": "both per-step exchanges (packed reduced camera system + step scalars), HIP events on the engine's stream in the "
                             "profiled (host-stepped) solve; 0 at one rank"},
        "pcie_inclusive": {"upload_ms": 1e3 * upload_s, "iters_per_sec": iters_done / (elapsed + upload_s),
                           "note": "host->device upload of the whole window (frames, points, descriptors, observations, cameras) + the same solve"},
        "device_copy_GBps": copy_gbps,
        "gen_seconds": t_gen,
    }

    def measure(eng_, prob_, steps, repeats, collective):
        """Median seconds per LM iteration of `repeats` timed solves of `steps` iterations on another engine / window
        (same bracket as the headline: barrier + synchronize on both sides, state reset outside)."""
        def reset():
            eng_.set_problem(prob_.xyz, prob_.desc, prob_.obs_point, prob_.obs_slot, prob_.weights)
            eng_.set_cameras(prob_.cams, prob_.fixed_slot)
        eng_.solve(opts(min(steps, 5)))
        ts = []
        for _ in range(max(1, repeats)):
            reset()
            torch.cuda.synchronize()
            if collective and dist is not None:
                dist.barrier()
            t1_ = time.perf_counter()
            raw_ = eng_.solve_raw(opts(steps), buffers=raw_buffers)
            torch.cuda.synchronize()
            if collective and dist is not None:
                dist.barrier()
            dt = time.perf_counter() - t1_
            n_it = len(Engine.unpack_solve(*raw_)["iterations"]) - 1
            if collective and dist is not None:
                t_ = torch.tensor([dt], dtype=torch.float64, device=ctl)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                dt = float(t_.item())
            ts.append(dt / max(1, n_it))
        ts.sort()
        return ts[len(ts) // 2]

    def plain_engine(prob_):
        _, rws, cls = prob_.images.shape
        e_ = Engine(rws, cls, prob_.K, prob_.radius, prob_.n_frames, huber=prob_.huber, device=local_rank)
        e_.load(prob_)
        return e_

    # ---- strong-scaling record of the SAME launch (N > 1, weak-scaling run): configs[3] by rank 0 alone, then sharded over the N ranks
    if world > 1 and not strong and os.environ.get("PBA_BENCH_STRONG", "1") != "0":
        try:
            n3 = int(os.environ.get("PBA_BENCH_STRONG_POINTS", "200000"))      # (tests shrink the window)
            whole3 = synthetic.make_window(n_frames=16, n_points=n3, radius=2, huber=0.0, dense_births=(0, 8))
            t1_full = None
            if rank == 0:
                e1 = plain_engine(whole3)
                t1_full = measure(e1, whole3, 20, 5, collective=False)
                e1.close()
            dist.barrier()
            sh3 = whole3.shard(rank, world)
            del whole3
            e3 = plain_engine(sh3)
            # the transport the headline engine ended up with (RCCL or host-staged, with or without the peer mailboxes) is the one
            # this engine uses: what failed its set-up or self-test there is not tried again inside a record
            if transport.startswith("RCCL"):
                uid3 = torch.zeros(128, dtype=torch.uint8, device=ctl)
                if rank == 0:
                    uid3.copy_(torch.frombuffer(bytearray(Engine.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid3, 0)
                e3.comm_init_rccl(bytes(uid3.cpu().numpy().tobytes()), rank, world)
            else:
                e3.comm_init_callback(_allreduce, rank, world)
            if eng.comm_transport().endswith("+peer"):
                e3.comm_enable_peer_exchange()
            tn = measure(e3, sh3, 20, 5, collective=True)
            tr3 = e3.comm_transport()
            e3.close()
            if rank == 0:
                out["strong"] = {"workload": "configs[3]: ONE 16-frame window, %d points (%d residual blocks), points sharded over the ranks" % (n3, 16 * n3),
                                 "us_per_iteration_1gpu": 1e6 * t1_full, "us_per_iteration": 1e6 * tn, "n_gpus": world,
                                 "speedup_vs_1gpu_same_run": t1_full / tn, "transport": tr3, "steps": 20, "repeats": 5}
        except Exception as exc:      # the headline line must survive a failure of the extra record
            if rank == 0:
                out["strong"] = {"error": repr(exc)}

    # ---- one rank of R on ONE GPU: the multi-rank code path at world = 1 and the projected strong-scaling speed-up ----------
    if emulate and world == 1:
        R = emulate
        os.environ["PBA_FORCE_MULTI"] = "1"          # read at pba_comm_init_*: the exchange path runs although world == 1
        sh = prob.shard(0, R)
        e2 = plain_engine(sh)
        e2.comm_init_rccl(Engine.comm_unique_id(), 0, 1)
        e2.comm_enable_peer_exchange()
        t_rank = measure(e2, sh, args.steps, max(3, args.repeats // 3), collective=False)
        tr2 = e2.comm_transport()
        # per-kernel shares of the rank's iteration: HIP events of the host-stepped driver (the async pipeline has no event mode)
        e2.set_problem(sh.xyz, sh.desc, sh.obs_point, sh.obs_slot, sh.weights)
        e2.set_cameras(sh.cams, sh.fixed_slot)
        e2.reset_counters()
        e2.solve(opts(min(args.steps, 10)))
        c2 = e2.counters()
        e2.close()
        del os.environ["PBA_FORCE_MULTI"]
        t_full = elapsed / max(1, iters_done)
        xg = 1e-6 * args.xgmi_exchange_us
        per = lambda k: (1e3 * c2[k + "_ms"] / c2[k + "_launches"]) if c2[k + "_launches"] else 0.0
        out["strong_projection"] = {
            "what": "configs[3] strong scaling projected from ONE GPU: the full window on the single-rank path, and rank 0's shard of %d "
                    "(%d points, %d residual blocks) on the MULTI-RANK path at world = 1 -- k_schur, k_reduce_final writing the mailbox slot + "
                    "flag, k_solve_blocked waiting for the flags and summing the slots, fused k_sample raising the flag, k_decide waiting and "
                    "summing -- so everything but the xGMI hop is measured" % (R, sh.n_points, sh.n_obs),
            "ranks": R, "us_per_iteration_full_window_1gpu": 1e6 * t_full, "us_per_iteration_one_rank": 1e6 * t_rank,
            "transport": tr2,
            "assumed_xgmi_us_per_exchange": args.xgmi_exchange_us, "exchanges_per_iteration": 2,
            "projected_us_per_iteration": 1e6 * (t_rank + 2 * xg),
            "projected_speedup": t_full / (t_rank + 2 * xg),
            "projected_speedup_if_exchange_free": t_full / t_rank,
            "bar": "north_star: >= 6x at 8 GPUs, i.e. <= %.1f us per iteration per rank" % (1e6 * t_full / 6.0),
            "rank_kernels_us_host_stepped": {"k_sample (fused)": per("linearize"), "k_schur": per("schur"),
                                             "k_reduce_final + k_solve_blocked (incl. its peer wait)": per("solve"),
                                             "scalar exchange kernel (host-stepped driver only)": per("exchange")},
        }

    # ---- CPU baseline: the oracle ("restated Ceres-equivalent CPU path") on a bounded sample, rank 0, N = 1 ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_steps > 0 and not args.inverse_depth:
        from oracle import oracle
        n_cpu = min(args.cpu_points, prob.n_points, max(1000, int(400000 / n_bar / args.channels)))     # bounded: <= 400k single-channel residual blocks
        sub = prob.shard(0, 1)
        hi = int(np.searchsorted(prob.obs_point, n_cpu, side="left"))
        sub.xyz, sub.desc = prob.xyz[:n_cpu].copy(), prob.desc[:n_cpu]
        sub.obs_point, sub.obs_slot = prob.obs_point[:hi], prob.obs_slot[:hi]
        threads = min(usable_cores(), 4)                 # reference default: min(omp_get_max_threads(), 4)
        o = oracle.default_options(max_num_iterations=args.cpu_steps, function_tolerance=0.0, gradient_tolerance=0.0,
                                   parameter_tolerance=0.0, num_threads=threads, use_autodiff=1)
        tc = time.perf_counter()
        cres = oracle.solve(sub, o)
        cpu_s = time.perf_counter() - tc
        it_cpu = len(cres["iterations"]) - 1
        frac = hi / float(n_obs_local)
        cpu_iters_per_sec_full = (it_cpu / cpu_s) * frac   # time scales linearly with the observation count
        out["cpu_baseline"] = {
            "value": cpu_iters_per_sec_full, "unit": "LM iters/s (extrapolated to the full %d-point window)" % prob.n_points,
            "cores": threads, "kind": "port",
            "sample": "first %d of %d points (%d observations), %d LM iterations, dual-number autodiff + materialised Jacobian "
                      "+ Schur, %d OpenMP threads, %.1f s wall" % (n_cpu, prob.n_points, hi, it_cpu, threads, cpu_s),
            "sample_iters_per_sec": it_cpu / cpu_s,
            "residuals_per_sec": hi * P * (cres["num_jacobian_passes"] + cres["num_cost_passes"]) / cpu_s,
        }
        out["speedup_vs_cpu_baseline"] = iters_per_sec / cpu_iters_per_sec_full
        # the same path at ALL host cores beside the reference's 4-thread cap (core count and CPU model stated)
        all_cores = usable_cores()
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        out["cpu_baseline"]["cpu_model"] = cpu_model
        out["cpu_baseline"]["host_cores"] = os.cpu_count() or 1
        out["cpu_baseline"]["usable_cores"] = all_cores          # affinity mask capped by the cgroup CPU quota
        if all_cores > threads and not args.no_cpu_all_cores:
            o.num_threads = all_cores
            o.max_num_iterations = args.cpu_steps
            tc = time.perf_counter()
            cres2 = oracle.solve(sub, o)
            cpu_s2 = time.perf_counter() - tc
            it2 = len(cres2["iterations"]) - 1
            out["cpu_baseline_all_cores"] = {
                "value": (it2 / cpu_s2) * frac, "unit": out["cpu_baseline"]["unit"], "cores": all_cores, "kind": "port", "cpu_model": cpu_model,
                "host_cores": os.cpu_count() or 1,
                "sample": "the same sample, %d LM iterations, %d OpenMP threads (every core the cgroup quota grants), %.1f s wall" % (it2, all_cores, cpu_s2), "sample_iters_per_sec": it2 / cpu_s2,
            }
            out["speedup_vs_cpu_all_cores"] = iters_per_sec / out["cpu_baseline_all_cores"]["value"]

    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
