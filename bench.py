#!/usr/bin/env python3
"""Benchmark of the ray-scene query path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One step = one frame of the workload rendered through the C ABI (rt_render): primary rays, closest
hits, area-light shadow rays (any-hit), reflection / refraction levels, with the scene resident in
HBM.  For N > 1 there is one rank per GPU (started by torch.distributed.run, or by this script itself
when it is called as plain `python bench.py --gpus N`): the image is cut into bands of 8 pixel rows
dealt round-robin to the ranks, the scene is replicated, and the bands are gathered to rank 0 over
RCCL inside the timed region (the gather of a frame overlaps the rendering of the next one; all of
them are complete before the clock stops); the total work is fixed ("strong" scaling).  With N > 1 a rank keeps
several frames in flight (rt_render_submit / rt_render_wait, FRAMES_IN_FLIGHT): its share of a frame is a short chain of
dependent launches that cannot fill a GPU by itself.  With N = 1 the timed loop is one blocking rt_render per frame (the
per-kernel times are then those of kernels running alone) and the rate with frames in flight is reported beside it
(`frames_in_flight`).

The W warm-up frames are followed by as many further untimed ones as it takes to have rendered for 50 ms
(RT_BENCH_PREWARM_S): the GPU reaches its clocks only after some twenty milliseconds of work, and a 1.6 ms frame takes
1.75 ms until then.  Then exactly K frames are timed.

Prints ONE JSON line on rank 0.  metric = Mrays/s, one ray = one Search-equivalent query of the
reference (primary, secondary or shadow).

`roofline` prices the dominant kernel (k_shadow: the any-hit pass over all area-light shadow rays)
against the FP32 vector peak in two ways: `achieved` / `frac` use the REFERENCE algorithm's work per ray
(3 + 18*B + 52*T flops with B slab tests and T ray/triangle tests per ray in the reference's
definitions, counted on the device by the instrumented pass and cross-checked by the CPU restatement on
a sample) -- a reference-equivalent rate; `executed_lane_ops_per_launch` / `valu_issue_frac` say what
the kernel really issues (VALU instruction counters of a rocprofv3 pass, committed under profiles/).
`cpu_baseline` is the CPU restatement on one host core, `cpu_baseline_all_cores` the same on all
cores of this box (threads over image rows), `cpu_baseline_reference` the unmodified reference binary.
`other_workloads` carries a few frames of the other benchmark configurations: with one GPU all of them, each with its
own one-core `cpu_baseline` on a few rows of its frame; with N > 1 the two the scaling target is stated on (the
1.31 M-triangle scene and the 4K fuzzysampling frame), band-split over the ranks like the main workload, gather included,
with the pipelined rate and the un-overlapped latency (first launch -> frame resident on rank 0) side by side.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Frames in flight run on streams of the library's own; the HIP runtime maps streams onto this many hardware queues
# (its default is 4, which serialises pairs of them).  Read when the runtime starts, so it is set before torch is loaded.
# (the GPU boxes export the default, so it is overridden, not defaulted; RT_BENCH_HW_QUEUES=0 leaves it alone)
if os.environ.get("RT_BENCH_HW_QUEUES", "8") != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_BENCH_HW_QUEUES", "8")

CAM_ORIGINAL = dict(cam_pos=[278.0, 273.0, -1000.0], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, 1.0], max_distance=2000.0)
CAM_SPHERE = dict(cam_pos=[0.0, 1.0, 4.42], cam_up=[0.0, 1.0, 0.0], cam_dir=[0.0, 0.0, -1.0], max_distance=6.0)

WORKLOADS = {
    # the configuration the headline metric is quoted on: Cornell-Original, Blinn-Phong, 16 samples per light, 1080p
    "cornell_original_blinn_spt16_1080p": dict(
        scene="cornell_original", cam=CAM_ORIGINAL, width=1920, height=1080, spp=1, spt=16, render_type="blinn_phong",
        shading="flat", cull=True, max_depth=10, cpu_rows=(0, 320)),
    # Cornell-Sphere (reflection + refraction), smooth shading, 4 samples per pixel, 32 samples per light
    "cornell_sphere_blinn_smooth_spp4_1080p": dict(
        scene="cornell_sphere", cam=CAM_SPHERE, width=1920, height=1080, spp=4, spt=32, render_type="blinn_phong",
        shading="smooth", cull=True, max_depth=10, cpu_rows=(540, 545)),
    # 1,310,754 triangles (Cornell box + icosphere level 8), hard shadows, true aspect ratio
    "synthetic_ico8_phong_4k": dict(
        scene="synthetic:8", cam=CAM_ORIGINAL, width=3840, height=2160, spp=1, spt=1, render_type="phong",
        shading="flat", cull=True, max_depth=10, aspect="true", cpu_rows=(1000, 1160), streams=3),
    # fuzzysampling (64 samples per pixel) + 32 samples per light, 4K
    "cornell_original_blinn_spp64_spt32_4k": dict(
        scene="cornell_original", cam=CAM_ORIGINAL, width=3840, height=2160, spp=64, spt=32, render_type="blinn_phong",
        shading="flat", cull=True, max_depth=10, cpu_rows=(1080, 1082), in_flight=2),
    # a small case for quick checks
    "cornell_original_blinn_spt4_256": dict(
        scene="cornell_original", cam=CAM_ORIGINAL, width=256, height=256, spp=1, spt=4, render_type="blinn_phong",
        shading="flat", cull=True, max_depth=10, cpu_rows=(96, 160)),
}
# "in_flight": frames in flight of the pipelined loops if not FRAMES_IN_FLIGHT (a frame of a second fills the GPU alone)
# "streams": sub-frames a rt_render call of the workload runs concurrently (rt_ads_set_render_streams); pays where a frame
# is a long chain of small recursion levels
DEFAULT_WORKLOAD = "cornell_original_blinn_spt16_1080p"
# shortest untimed warm-up of a frame loop, seconds (FrameLoop.warm)
PREWARM_S = float(os.environ.get("RT_BENCH_PREWARM_S", "0.05"))
# frames of the other configurations that the default single-GPU run appends to its line
OTHER_WORKLOADS = {"cornell_sphere_blinn_smooth_spp4_1080p": 4, "synthetic_ico8_phong_4k": 5, "cornell_original_blinn_spp64_spt32_4k": 2}
# ... and the ones a multi-GPU run appends (band-split over the ranks): the workloads the scaling target names
OTHER_WORKLOADS_MULTI = {"synthetic_ico8_phong_4k": 6, "cornell_original_blinn_spp64_spt32_4k": 2}

# Frames in flight (rt_render_submit / rt_render_wait) of the pipelined loops: with N > 1 for `value` (a rank's share of a
# frame is a short chain of launches that cannot fill the GPU alone), with N = 1 reported beside the blocking loop.
# Measured on one MI355X with 8 hardware queues: 4 is best or within 1 % of it for every workload and split.
FRAMES_IN_FLIGHT = int(os.environ.get("RT_BENCH_IN_FLIGHT") or 4)



def depth_for(wl):
    return min(FRAMES_IN_FLIGHT, wl.get("in_flight", FRAMES_IN_FLIGHT))


FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X: 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2.4 GHz
FP32_NO_FMA_PEAK_TOPS = 78.6      # the same lanes issuing one non-fused op per clock (bit-parity forbids FMA)
HBM_PEAK_GBPS = 8000.0


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run as a child
    process, before this process has touched torch or HIP) and pass rank 0's JSON line through."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def load_scene(wl, Scene):
    if wl["scene"].startswith("synthetic:"):
        from raytracer_amd.synthetic import cornell_with_icosphere
        box = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_original.npz"))
        return cornell_with_icosphere(box, int(wl["scene"].split(":")[1]))
    return Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", wl["scene"] + ".npz"))


def render_kwargs(wl):
    aspect = 1 if wl.get("aspect") == "true" else 0
    return dict(fov=0.6, width=wl["width"], height=wl["height"], spp=wl["spp"], spt=wl["spt"], render_type=wl["render_type"],
                shading=wl["shading"], cull=wl["cull"], max_depth=wl["max_depth"], seed=42, aspect_mode=aspect, **wl["cam"])


def reference_work(ads, rt, wl, max_rays=6e7):
    """B (slab tests) and T (ray/triangle tests) per ray in the REFERENCE's definitions, for any-hit and closest-hit
    rays separately, counted on the device by the instrumented pass (rt_ads_collect_search_stats) on this workload's
    frame -- at a reduced image size if the full frame has more than max_rays rays (per-ray averages do not depend on
    the size; samples per pixel are reduced to 1 for the same reason)."""
    kw = render_kwargs(wl)
    kw["spp"] = 1
    est = wl["width"] * wl["height"] * (1.0 + 2.2 * wl["spt"])
    k = 1
    while est / (k * k) > max_rays:
        k += 1
    kw["width"], kw["height"] = max(64, wl["width"] // k), max(64, wl["height"] // k)
    if wl.get("aspect") != "true" and wl["height"] < wl["width"]:
        kw["height"] = min(kw["height"], 64)  # compat aspect with width > height: every row is the same scanline
    ads.reset_query_stats()
    ads.collect_search_stats(True)
    ads.render(rt.make_render_config(**kw))
    ads.collect_search_stats(False)
    q = ads.query_stats()
    n_any = max(q["counted_rays_any"], 1)
    n_cl = max(q["counted_rays"] - q["counted_rays_any"], 1)
    b_any, t_any = q["slab_tests_any"] / n_any, q["ray_triangle_tests_any"] / n_any
    b_cl = (q["slab_tests"] - q["slab_tests_any"]) / n_cl
    t_cl = (q["ray_triangle_tests"] - q["ray_triangle_tests_any"]) / n_cl
    return {"any_hit": {"B": b_any, "T": t_any, "flops": 3 + 18 * b_any + 52 * t_any},
            "closest_hit": {"B": b_cl, "T": t_cl, "flops": 3 + 18 * b_cl + 52 * t_cl},
            "counted_on": f"device, instrumented pass, {kw['width']}x{kw['height']} frame of the same workload, {q['counted_rays']} rays"}


def committed_reference_work(workload):
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_bench_n1.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.loads(f.read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError):
            continue
        w = d.get("reference_work_per_ray") if d.get("config", {}).get("workload") == workload else \
            d.get("other_workloads", {}).get(workload, {}).get("reference_work_per_ray")
        if w and "counted_on" in w and not w["counted_on"].startswith("quoted"):
            w = {k: v for k, v in w.items() if k != "cpu_cross_check"}
            w["counted_on"] = f"quoted from profiles/{os.path.basename(path)}: " + w["counted_on"]
            return w
    return None


def kernel_summary(acc, steps, work):
    """Per-kernel-class rates from the HIP-event times that rt_render reports (first 64 launches of each class per frame)."""
    shadow_rate = acc["shadow_rays"] / max(acc["shadow_ms"] * 1e-3, 1e-12)
    closest_rate = acc["closest_rays"] / max(acc["closest_ms"] * 1e-3, 1e-12)
    out = {}
    for name, key in (("k_shadow", "shadow"), ("k_closest", "closest"), ("k_shade", "shade")):
        timed, every = acc[key + "_launches"], acc[key + "_launches_all"]
        avg = acc[key + "_ms"] / max(timed, 1)
        out[name] = {"launches": timed, "avg_launch_ms": avg, "ms_per_frame": avg * every / steps}
        if every != timed:  # frames with more than 64 launches of the class: the untimed ones are taken at the timed average
            out[name]["launches_untimed"] = every - timed
    if acc["tail_launches"]:
        # the fused launch that runs what is left of the recursion once a level is small (closest hit, shadow rays and
        # shading of every further level)
        out["k_tail"] = {"launches": acc["tail_launches"], "avg_launch_ms": acc["tail_ms"] / acc["tail_launches"],
                         "ms_per_frame": acc["tail_ms"] / steps}
    out["k_shadow"]["Mrays_per_s"] = shadow_rate / 1e6
    if acc["shadow_rays_not_traced"]:
        # shadow rays (counted like the reference casts them) whose light sample adds a zero to the colour whether it is
        # visible or not: answered without a walk (rt_render_stats.shadow_rays_not_traced)
        out["k_shadow"]["rays_not_traced_frac"] = acc["shadow_rays_not_traced"] / max(acc["all_shadow_rays"], 1)
    out["k_closest"]["Mrays_per_s"] = closest_rate / 1e6
    if work is not None:
        for name, rate, key in (("k_shadow", shadow_rate, "any_hit"), ("k_closest", closest_rate, "closest_hit")):
            tf = rate * work[key]["flops"] / 1e12
            out[name]["reference_equivalent_TFLOPs"] = tf
            out[name]["frac_of_fp32_vector_peak"] = tf / FP32_VECTOR_PEAK_TFLOPS
            out[name]["frac_of_no_fma_issue_peak"] = tf / FP32_NO_FMA_PEAK_TOPS
    return out


def new_acc():
    return dict(tail_ms=0.0, tail_launches=0, rays=0, shade_ms=0.0, closest_ms=0.0, shadow_ms=0.0, shade_launches=0, closest_launches=0, shadow_launches=0,
                shade_launches_all=0, closest_launches_all=0, shadow_launches_all=0, shadow_rays=0, closest_rays=0, all_shadow_rays=0,
                shadow_rays_not_traced=0)


def add_stats(acc, st):
    acc["rays"] += st.rays
    acc["shade_ms"] += st.ms_shade
    acc["closest_ms"] += st.ms_closest
    acc["shadow_ms"] += st.ms_shadow
    # per-kernel times come from HIP events around the first 64 launches of each class in a call; the ray counts
    # used for kernel rates are those of exactly these launches
    acc["shade_launches"] += min(st.launches_shade, 64)
    acc["shadow_launches"] += min(st.launches_shadow, 64)
    acc["closest_launches"] += min(st.launches_closest, 64)
    acc["tail_ms"] += st.ms_tail
    acc["tail_launches"] += st.launches_tail
    acc["shade_launches_all"] += st.launches_shade
    acc["shadow_launches_all"] += st.launches_shadow
    acc["closest_launches_all"] += st.launches_closest
    acc["shadow_rays"] += st.timed_shadow_rays
    acc["closest_rays"] += st.timed_closest_rays
    acc["all_shadow_rays"] += st.shadow_rays
    acc["shadow_rays_not_traced"] += st.shadow_rays_not_traced


def pmc_file():
    """The counter summary to quote: the committed one of the latest round, or the file RT_BENCH_COUNTERS names (a
    summary just collected on this box, tools/refresh_round.sh)."""
    import glob
    override = os.environ.get("RT_BENCH_COUNTERS")
    if override and os.path.exists(override):
        return override, os.path.relpath(os.path.abspath(override), ROOT) + " (collected on this box just before this run; not a committed file)"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.json")))
    return (files[-1], "profiles/" + os.path.basename(files[-1])) if files else (None, None)


def pmc_summary():
    """The per-workload counters of pmc_file() (tools/pmc_profile.sh), or {}."""
    path, src = pmc_file()
    if not path:
        return {}, None
    with open(path) as f:
        return json.load(f).get("workloads", {}), src


def memory_side(kernels, name, workload):
    """Beyond-L2 traffic of the two traversal kernels from the counter summary, priced with this run's kernel times:
    on a tree far larger than the L2 the per-lane any-hit pass is bound by scattered 64..128-byte reads, not by issue."""
    pmc, src = pmc_summary()
    for k in ("k_shadow", "k_closest"):
        e = pmc.get(workload, {}).get("kernels", {}).get(k)
        if not e or "bytes_per_launch" not in e or not kernels[k]["avg_launch_ms"]:
            continue
        gbps = e["bytes_per_launch"] / (kernels[k]["avg_launch_ms"] * 1e-3) / 1e9
        kernels[k]["beyond_l2_GBps"] = gbps
        kernels[k]["hbm_frac"] = gbps / HBM_PEAK_GBPS
        kernels[k]["l2_hit_rate"] = e.get("l2_hit_rate")
        if e.get("valu_lane_ops_per_launch"):
            # executed wave64 VALU lane-ops (counted, per-launch mean) over this run's kernel time and the no-FMA issue peak
            kernels[k]["valu_issue_frac"] = e["valu_lane_ops_per_launch"] / (kernels[k]["avg_launch_ms"] * 1e-3) / 1e12 / FP32_NO_FMA_PEAK_TOPS
            kernels[k]["lanes_active_per_valu_inst"] = e.get("lanes_active")
        kernels[k]["counters_source"] = src


def build_ads(rt, scene, wl, device=0):
    ads = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10}, device=device)
    ads.Build(scene.pos)
    s0 = time.perf_counter()
    ads.set_shading(scene.norm, scene.matid, scene.materials)
    # (materials, lights and the per-light lists / trees of the any-hit passes: host work, once per scene)
    ads.shading_setup_s = time.perf_counter() - s0
    ads.set_render_streams(int(os.environ.get("RT_BENCH_STREAMS") or wl.get("streams", 1)))
    return ads


class FrameLoop:
    """Renders frames of one workload on this rank's bands and moves them to rank 0.

    A ring of frame buffers: while the bands of one frame travel to rank 0 (RCCL, on its own stream), the next frame is
    rendered into another buffer.  overlap=False completes every gather before the next frame starts.  The gathers
    run on a worker thread with its own stream (use_thread=False: on the calling thread, between the frames): the
    renderer's call only returns when its frame is complete, so whatever the calling thread does between two calls --
    the host side of the collective included -- would otherwise add to every frame.

    depth > 0: that many frames are in flight on the GPU (rt_render_submit / rt_render_wait); a frame is handed to the
    gather when it is retired.  depth = 0: one blocking rt_render call per frame."""

    def __init__(self, ads, rt, wl, rank, world, dist, torch, overlap=True, use_thread=True, depth=0, device="cuda"):
        from raytracer_amd import distributed as rd
        self.ads, self.torch, self.dist = ads, torch, dist
        self.device = device
        self.on_gpu = str(device).startswith("cuda")  # ("cpu": the loop's bookkeeping under test, with gloo and a stand-in renderer)
        self.H, self.W = wl["height"], wl["width"]
        self.cfg = rt.make_render_config(band_rank=rank, band_world=world, **render_kwargs(wl))
        self.depth = depth if overlap or dist is None else 0
        nbuf = 2 if self.depth == 0 else self.depth + 2
        self.frames = [rd.alloc_frame(self.H, self.W, world, device) for _ in range(nbuf)]
        self.stream = torch.cuda.current_stream().cuda_stream if self.on_gpu else None
        self.gathers = [rd.BandGather(self.H, self.W, rank, world, device, dist) for _ in range(nbuf)]
        self.overlap = dist is not None and overlap
        self.worker = rd.GatherThread(self.frames, self.gathers, device=device) if self.overlap and use_thread else None
        self.n = 0
        self.in_flight = []  # (ticket, buffer) of the frames submitted and not yet retired, oldest first

    def _claim(self):
        """The next buffer of the ring, once its previous frame has arrived on rank 0."""
        k = self.n % len(self.frames)
        self.n += 1
        if self.worker is not None:
            self.worker.wait(k)
        else:
            self.gathers[k].finish()
        return k

    def _hand_over(self, k):
        """Buffer k holds a complete frame: its bands start travelling to rank 0."""
        if self.worker is not None:
            self.worker.submit(k)
            return
        self.gathers[k].start(self.frames[k])  # no-op for one rank
        if not self.overlap:
            self.gathers[k].finish()

    def _retire(self):
        ticket, k = self.in_flight.pop(0)
        st = self.ads.render_wait(ticket)
        self._hand_over(k)
        return st

    def step(self):
        """One frame, complete on return (its gather may still travel)."""
        k = self._claim()
        if self.depth == 0:
            _, st = self.ads.render(self.cfg, out_ptr=self.frames[k].data_ptr(), stream=self.stream)
            self._hand_over(k)
            return st
        self.in_flight.append((self.ads.render_submit(self.cfg, self.frames[k].data_ptr(), self.stream), k))
        return self._retire()

    def fence(self):
        if self.worker is not None:
            self.worker.wait()
        for g_ in self.gathers:
            g_.finish()
        if self.on_gpu:
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def timed(self, steps, acc=None):
        """Seconds for `steps` frames between two fences (barrier + device synchronisation on both sides)."""
        self.fence()
        t0 = time.perf_counter()
        if self.depth == 0:
            for _ in range(steps):
                st = self.step()
                if acc is not None:
                    add_stats(acc, st)
        else:
            for i in range(steps + self.depth):
                if len(self.in_flight) == self.depth or i >= steps:
                    if not self.in_flight:
                        break
                    st = self._retire()
                    if acc is not None:
                        add_stats(acc, st)
                if i < steps:
                    k = self._claim()
                    self.in_flight.append((self.ads.render_submit(self.cfg, self.frames[k].data_ptr(), self.stream), k))
        self.fence()
        return time.perf_counter() - t0

    def warm(self, frames=1):
        """Untimed frames through the same loop.  Every frame slot of the library allocates its level buffers on its first
        frame, so a loop with frames in flight is warm only after each slot has rendered one.  The GPU also needs some
        twenty milliseconds of work to reach its clocks (a 1.6 ms frame takes 1.75 ms in the first ten): the warm-up lasts
        at least PREWARM_S, with a number of further frames that every rank derives from the same (slowest) rank's time."""
        n = max(frames, 2 * self.depth)
        spent = max_over_ranks(self.timed(n), self.dist, self.torch, device=self.device)
        if spent < PREWARM_S:
            more = min(int((PREWARM_S - spent) / max(spent / n, 1e-6)) + 1, 400)
            self.timed(max(more, self.depth))

    def last_frame(self):
        return self.frames[(self.n - 1) % len(self.frames)]

    def close(self):
        while self.in_flight:
            self._retire()
        if self.worker is not None:
            self.worker.close()
            self.worker = None


def max_over_ranks(seconds, dist, torch, device="cuda"):
    if dist is None:
        return seconds
    tt = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def sum_over_ranks(count, dist, torch):
    if dist is None:
        return count
    rr = torch.tensor([count], dtype=torch.int64, device="cuda")
    dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    return int(rr.item())


def cpu_sample(wl, scene, frame_rows=None):
    """The CPU restatement (oracle/rt_oracle.c, one host core) on the workload's cpu_rows: its rate, and -- if the
    device frame's rows are given -- the largest difference between the two on those rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc
    osc = orc.Scene.from_arrays(scene.pos, scene.norm, scene.matid, scene.materials)
    b0 = time.perf_counter()
    otree = orc.Octree(osc, 16, 10)
    build_s = time.perf_counter() - b0
    r0, r1 = wl["cpu_rows"]
    ocfg = orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=(r0, r1), **render_kwargs(wl))
    c0 = time.perf_counter()
    ref, _, _, ost = orc.render(osc, otree, ocfg, want_hits=False)
    cpu_s = time.perf_counter() - c0
    out = {"value": ost.search_calls / cpu_s / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
           "sample": f"rows {r0}..{r1 - 1} of the same frame ({ost.search_calls} rays, {cpu_s:.1f} s; octree built in {build_s:.1f} s, not "
                     "counted), oracle/rt_oracle.c -O2, counter RNG"}
    parity = None
    if frame_rows is not None:
        diff = float(np.abs(frame_rows - ref[r0:r1]).max())
        parity = {"max_abs_diff": diff, "tolerance": 1.0 / 255.0, "ok": diff <= 1.0 / 255.0}
    return out, parity, (osc, otree, ost, orc)


def run_other_workload(name, frames, rt, Scene, torch, rank=0, world=1, dist=None, cpu=True):
    """A few frames of another benchmark configuration (device-resident frame buffer): on one GPU with its kernel-class
    rates, reference work and a one-core CPU sample; on N ranks band-split with the gather, pipelined and un-overlapped."""
    wl = WORKLOADS[name]
    scene = load_scene(wl, Scene)
    ads = build_ads(rt, scene, wl, device=torch.cuda.current_device())
    loop = FrameLoop(ads, rt, wl, rank, world, dist, torch, depth=depth_for(wl) if dist is not None else 0)
    loop.warm()
    acc = new_acc()
    elapsed = max_over_ranks(loop.timed(frames, acc), dist, torch)
    rays = sum_over_ranks(acc["rays"], dist, torch)
    loop.close()
    ts = ads.tree_stats()
    res = {"Mrays_per_s": rays / elapsed / 1e6, "ms_per_frame": elapsed / frames * 1e3, "frames": frames,
           "rays_per_frame": rays // frames, "triangles": scene.n_triangles, "octree_nodes": ts["nodes"],
           "octree_build_s": round(ts["build_seconds"], 4), "shading_setup_s": round(ads.shading_setup_s, 4), "render_streams": int(os.environ.get("RT_BENCH_STREAMS") or wl.get("streams", 1)),
           "n_gpus": world}
    if dist is not None:
        res["frames_in_flight"] = loop.depth
        # the same frames one after the other: rendering, gather and interleaving copy of a frame complete before the next
        # one starts -- the latency from the first launch to the frame resident on rank 0
        sync = FrameLoop(ads, rt, wl, rank, world, dist, torch, overlap=False)
        sync.step()
        n_sync = max(2, frames // 2)
        res["unoverlapped_ms_per_frame"] = max_over_ranks(sync.timed(n_sync), dist, torch) / n_sync * 1e3
        res["pipelined_ms_per_frame"] = res["ms_per_frame"]
        sync.close()
    else:
        # kernel-class times: with concurrent sub-frames the launches of a class overlap one another, so the classes
        # are timed on frames rendered as one piece
        if res["render_streams"] > 1:
            ads.set_render_streams(1)
            one = FrameLoop(ads, rt, wl, rank, world, dist, torch)
            one.step()
            acc = new_acc()
            n_one = max(2, frames // 2)
            res["ms_per_frame_one_stream"] = one.timed(n_one, acc) / n_one * 1e3
            frames_k = n_one
            one.close()
        else:
            frames_k = frames
        # the same frames with several in flight (what a multi-GPU run uses for its shares)
        if FRAMES_IN_FLIGHT > 0 and res["ms_per_frame"] < 200.0:  # (a frame of hundreds of milliseconds fills the GPU by itself)
            ads.set_render_streams(1)
            fl = FrameLoop(ads, rt, wl, rank, world, dist, torch, depth=FRAMES_IN_FLIGHT)
            fl.warm()
            facc = new_acc()
            n_fl = max(frames, 2 * FRAMES_IN_FLIGHT, min(40, int(100.0 / res["ms_per_frame"])))  # (filling and draining the pipeline are inside)
            fl_s = fl.timed(n_fl, facc)
            fl.close()
            res["in_flight"] = {"depth": FRAMES_IN_FLIGHT, "frames": n_fl, "ms_per_frame": fl_s / n_fl * 1e3, "Mrays_per_s": facc["rays"] / fl_s / 1e6}
        work = reference_work(ads, rt, wl)
        ks = kernel_summary(acc, frames_k, work)
        memory_side(ks, name, name)
        res.update({"reference_work_per_ray": work, "kernels": ks,
                    "k_shadow_frac": ks["k_shadow"].get("frac_of_fp32_vector_peak"), "k_closest_frac": ks["k_closest"].get("frac_of_fp32_vector_peak"),
                    "k_shadow_hbm_frac": ks["k_shadow"].get("hbm_frac")})
        if cpu:
            r0, r1 = wl["cpu_rows"]
            rows = loop.last_frame()[r0:r1].cpu().numpy()
            res["cpu_baseline"], res["parity_on_sample"], _ = cpu_sample(wl, scene, rows)
            res["gpu_over_cpu_one_core"] = res["Mrays_per_s"] / res["cpu_baseline"]["value"]
    ads.close()
    del loop
    return res


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    n = int(os.environ.get("OMP_NUM_THREADS") or 0) or 16
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = min(n, os.cpu_count() or 1)
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed frames: no reference-work pass, no other workloads, no host-framebuffer timing (counter passes)")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="frames in flight of the timed loop (rt_render_submit / rt_render_wait); default: 0 = one blocking rt_render per "
                         "frame with one GPU, %d with more" % FRAMES_IN_FLIGHT)
    args = ap.parse_args()

    # RT_BENCH_SELF_LAUNCH=1 takes the self-launch path with one rank too (a check of it on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("RT_BENCH_SELF_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    # stdout carries exactly one JSON line.  Native libraries (RCCL prints a version banner when a communicator is
    # created) write to file descriptor 1 directly, so everything but the final line is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import raytracer_amd as rt
    from raytracer_amd.scene import Scene

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dist = None
    # RT_BENCH_FORCE_DIST=1 runs the RCCL path (gather, reductions, barriers) with a single rank: a check of the
    # multi-rank code on a one-GPU box.
    force_dist = world == 1 and os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    scene = load_scene(wl, Scene)
    ads = build_ads(rt, scene, wl, device=local_rank)
    tree_stats = ads.tree_stats()
    W, H = wl["width"], wl["height"]
    kw = render_kwargs(wl)
    aspect = kw["aspect_mode"]
    cfg = rt.make_render_config(band_rank=rank, band_world=world, **kw)

    depth = args.frames_in_flight if args.frames_in_flight is not None else (depth_for(wl) if dist is not None else 0)
    if depth < 0 or depth > 8:
        raise SystemExit("--frames-in-flight must be in 0..8")
    loop = FrameLoop(ads, rt, wl, rank, world, dist, torch, overlap=os.environ.get("RT_BENCH_SYNC_GATHER") != "1",
                     use_thread=os.environ.get("RT_BENCH_GATHER_THREAD") != "0", depth=depth)
    depth = loop.depth
    loop.warm(args.warmup)
    acc = new_acc()
    elapsed = max_over_ranks(loop.timed(args.steps, acc), dist, torch)
    rays = sum_over_ranks(acc["rays"], dist, torch)
    frame = loop.last_frame()
    loop.close()
    in_flight = None
    if world == 1 and dist is None and depth == 0 and FRAMES_IN_FLIGHT > 0 and not args.no_extras:
        # the same frames with several in flight: what a sequence of frames renders at when the caller does not wait for
        # each one (reported beside `value`, which stays one blocking call per frame so that the per-kernel times of the
        # timed region are those of kernels running alone)
        fl = FrameLoop(ads, rt, wl, rank, world, dist, torch, depth=FRAMES_IN_FLIGHT)
        fl.warm(args.warmup)
        facc = new_acc()
        fl_s = fl.timed(args.steps, facc)
        fl.close()
        in_flight = {"depth": FRAMES_IN_FLIGHT, "frames": args.steps, "ms_per_frame": fl_s / args.steps * 1e3,
                     "Mrays_per_s": facc["rays"] / fl_s / 1e6,
                     "note": "rt_render_submit / rt_render_wait: the same frames, bit-identical images, %d in flight" % FRAMES_IN_FLIGHT}
    unoverlapped_ms = None
    if dist is not None:
        sync = FrameLoop(ads, rt, wl, rank, world, dist, torch, overlap=False)
        sync.step()
        n_sync = max(2, min(args.steps, 10))
        unoverlapped_ms = max_over_ranks(sync.timed(n_sync), dist, torch) / n_sync * 1e3
        sync.close()

    others_multi = None
    # (RT_BENCH_FORCE_DIST=1 RT_BENCH_MULTI_OTHERS=1 runs this path with a single rank: a check of it on a one-GPU box)
    multi = world > 1 or os.environ.get("RT_BENCH_MULTI_OTHERS") == "1"
    if dist is not None and multi and not args.no_extras and not args.no_other_workloads and args.workload == DEFAULT_WORKLOAD:
        # every rank takes part: the frames are band-split and gathered like the main workload's
        others_multi = {name: run_other_workload(name, n_frames, rt, Scene, torch, rank, world, dist, cpu=False)
                        for name, n_frames in OTHER_WORKLOADS_MULTI.items()}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary+secondary+shadow Search-equivalent queries)",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_frame": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic camera over the Cornell-box scene dump (tests/golden/scenes); counter-based RNG",
            "config": {"workload": args.workload, "width": W, "height": H, "samples_per_pixel": wl["spp"],
                       "samples_per_triangle": wl["spt"], "render_type": wl["render_type"], "shading": wl["shading"],
                       "octree": "max_triangles_per_BB=16,max_depth=10", "aspect_mode": "true" if aspect else "compat",
                       "triangles": scene.n_triangles, "octree_nodes": tree_stats["nodes"], "octree_build_s": round(tree_stats["build_seconds"], 3), "shading_setup_s": round(ads.shading_setup_s, 4),
                       "rays_per_frame": rays // args.steps, "partition": f"8-row bands round-robin over {world} rank(s)",
                       "frames_in_flight": depth},
        }
        if in_flight is not None:
            out["frames_in_flight"] = in_flight
        if unoverlapped_ms is not None:
            # `value` is the pipelined rate (the gather of a frame overlaps the rendering of the next); this is one frame
            # from its first launch to the framebuffer resident on rank 0, nothing overlapped
            out["frame_latency"] = {"pipelined_ms_per_frame": ms_per_step, "unoverlapped_ms_per_frame": unoverlapped_ms,
                                    "note": "un-overlapped: rendering, RCCL gather of the bands and the interleaving copy of a frame "
                                            "complete before the next frame starts"}
        work = None
        if world == 1 and not args.no_extras:
            # reference work per ray, counted on the device in the reference's definitions (whole reduced frame)
            work = reference_work(ads, rt, wl)
            out["reference_work_per_ray"] = work
        elif world == 1:
            # a run without extra launches (the one that is profiled): the per-ray work is a constant of the workload --
            # same scene, camera and generator -- and is quoted from the newest committed default run
            work = committed_reference_work(args.workload)
            if work is not None:
                out["reference_work_per_ray"] = work
        # ---- CPU baselines on bounded samples of the same frame (rank 0, N = 1 only) ----
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            r0, r1 = wl["cpu_rows"]
            okw = {k: v for k, v in kw.items()}
            model = cpu_model()
            out["cpu_baseline"], out["parity_on_sample"], (osc, otree, ost, orc) = cpu_sample(wl, scene, frame[r0:r1].cpu().numpy())
            # the CPU restatement's own count of the reference work on the sampled rows: cross-check of the device counters
            n_any = max(ost.shadow_rays, 1)
            n_closest = max(ost.primary_rays + ost.secondary_rays, 1)
            out["reference_work_per_ray"]["cpu_cross_check"] = {
                "any_hit": {"B": ost.slab_tests_any / n_any, "T": ost.tri_tests_any / n_any},
                "closest_hit": {"B": (ost.slab_tests - ost.slab_tests_any) / n_closest, "T": (ost.tri_tests - ost.tri_tests_any) / n_closest},
                "counted_on": f"oracle, rows {r0}..{r1 - 1}"}
            # all host cores of this box: threads over row blocks of the same restatement (the C code releases the GIL;
            # the counter generator makes rows independent), sized to about 16x the single-core sample
            threads = host_threads()
            rows_total = min(H, (r1 - r0) * max(threads, 1))
            a0 = max(0, min(r0, H - rows_total))
            edges = [a0 + (rows_total * i) // (threads * 4) for i in range(threads * 4 + 1)]
            jobs = [(edges[i], edges[i + 1]) for i in range(threads * 4) if edges[i + 1] > edges[i]]
            from concurrent.futures import ThreadPoolExecutor

            def job(rows):
                _, _, _, s_ = orc.render(osc, otree, orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=rows, **okw), want_hits=False)
                return s_.search_calls

            c0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=threads) as ex:
                all_rays = sum(ex.map(job, jobs))
            all_s = time.perf_counter() - c0
            out["cpu_baseline_all_cores"] = {"value": all_rays / all_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "cpu": model,
                                             "sample": f"rows {a0}..{a0 + rows_total - 1} of the same frame in {len(jobs)} row blocks over "
                                                       f"{threads} threads ({all_rays} rays, {all_s:.1f} s), oracle/rt_oracle.c -O2, counter RNG"}
            # The unmodified reference binary itself (built from the reference's sources where they exist; the binary
            # travels with the repository), on the same scene and settings at a reduced image size, one sample per
            # pixel: about 5 M rays.  The time is the whole process (scene load, Build, frame, PPM write).
            import ref_runner
            if ref_runner.available() and not wl["scene"].startswith("synthetic:"):
                scale = max(1, int(round(((rays / args.steps / wl["spp"]) / 5e6) ** 0.5)))
                rw, rh = max(8, W // scale), max(8, H // scale)
                rr = ref_runner.run(scene.pos, scene.norm, scene.matid, scene.materials, wl["cam"]["cam_pos"], wl["cam"]["cam_up"],
                                    wl["cam"]["cam_dir"], 0.6, rw, rh, spp=1, spt=wl["spt"], render_type=wl["render_type"],
                                    shading=wl["shading"], cull=wl["cull"], max_depth=wl["max_depth"],
                                    max_distance=wl["cam"]["max_distance"])
                out["cpu_baseline_reference"] = {
                    "value": rr["search_calls"] / rr["wall_s"] / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference", "cpu": model,
                    "sample": f"oracle/_ref/raytracer_ref (-O2, glibc rand) on the same scene and settings at {rw}x{rh}, 1 sample "
                              f"per pixel: {rr['search_calls']} Search calls in {rr['wall_s']:.1f} s (whole process)"}
        if world == 1 and not args.no_extras:
            # the same frame with the framebuffer handed back in host memory (device -> host copy inside the call);
            # reported beside `value`, never as it
            host_img = np.empty((H, W, 3), np.float32)
            ads.render(cfg, out=host_img)
            h0 = time.perf_counter()
            n_host = max(1, min(args.steps, 5))
            for _ in range(n_host):
                _, hst = ads.render(cfg, out=host_img)
            host_s = (time.perf_counter() - h0) / n_host
            out["host_framebuffer"] = {"Mrays_per_s": hst.rays / host_s / 1e6, "ms_per_frame": host_s * 1e3,
                                       "note": "rt_render with RT_MEM_HOST output: adds the PCIe copy of the frame"}
        ks = kernel_summary(acc, args.steps, work)
        sh = ks["k_shadow"]
        roof = {"bound": "valu", "kernel": "k_shadow (any-hit pass: area-light shadow rays)",
                "achieved": sh.get("reference_equivalent_TFLOPs"), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": sh.get("frac_of_fp32_vector_peak"), "traffic": None,
                "achieved_is": "reference-equivalent work: rays x (3 + 18 B + 52 T) flops of the reference algorithm per ray / kernel time; "
                               "the kernel executes less (shared plane products, early exits, no duplicate triangles, and with back-face culling "
                               "the triangles that can never occlude the light are left out of its lists) and, bit-exact without FMA, can "
                               "issue at most 78.6 T lane-ops/s -- so this fraction may exceed what the no-FMA issue peak would allow the "
                               "reference's own operation count; valu_issue_frac is what the kernel really issues",
                "frac_of_no_fma_issue_peak": sh.get("frac_of_no_fma_issue_peak"),
                "algorithmic_flops_per_ray": work["any_hit"]["flops"] if work else None,
                "launches": sh["launches"], "avg_launch_ms": sh["avg_launch_ms"],
                "rays_per_launch": acc["shadow_rays"] / max(acc["shadow_launches"], 1), "kernel_Mrays_per_s": sh["Mrays_per_s"],
                "other_kernels": {"k_closest": ks["k_closest"], "k_shade": ks["k_shade"]}}
        # Counter evidence comes from separate rocprofv3 --pmc passes of this same command (tools/pmc_profile.sh); they
        # cannot be collected from inside the run, so the committed summary of the latest round is quoted.
        pmc_path, src = pmc_file()
        if pmc_path and world == 1:
            with open(pmc_path) as f:
                pmc = json.load(f)
            k = pmc.get("workloads", {}).get(args.workload, {}).get("kernels", {}).get("k_shadow")
            if k:
                if "bytes_per_launch" in k:
                    roof["traffic"] = k["bytes_per_launch"]
                    roof["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, reads doubled per the gfx950 rule; bytes per launch)"
                    n_lights = max(len(scene.light_triangles()), 1)
                    hits = (acc["shadow_rays"] / max(acc["shadow_launches"], 1)) / (wl["spt"] * n_lights)
                    # per hit: work index + t + triangle index read (12 B), one visibility word per light and 32 samples written
                    roof["algorithmic_bytes_per_launch"] = hits * (12.0 + 4.0 * n_lights * ((wl["spt"] + 31) // 32))
                    roof["hbm_frac"] = roof["traffic"] / (sh["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
                if "valu_lane_ops_per_launch" in k:
                    roof["executed_lane_ops_per_launch"] = k["valu_lane_ops_per_launch"]
                    roof["executed_valu_insts_per_launch"] = k.get("SQ_INSTS_VALU")
                    roof["executed_salu_insts_per_launch"] = k.get("SQ_INSTS_SALU")
                    roof["valu_issue_frac"] = k["valu_lane_ops_per_launch"] / (sh["avg_launch_ms"] * 1e-3) / 1e12 / FP32_NO_FMA_PEAK_TOPS
                    roof["lanes_active_per_valu_inst"] = k.get("lanes_active")
                    roof["lds_bank_conflict_frac"] = k.get("lds_bank_conflict_frac")
                    roof["counters_source"] = src + " (rocprofv3 --pmc SQ_* passes, per-launch means; valu_issue_frac divides the executed wave64 " \
                                                    "VALU lane-ops by this run's kernel time and the 78.6 T lane-ops/s issue peak)"
        if depth > 0:
            roof["kernel_times_note"] = ("frames in flight overlap: a kernel's event interval includes the kernels of other frames that ran "
                                         "beside it, so these per-kernel figures are lower bounds of the kernels' own rates (the N = 1 run "
                                         "times its kernels one frame at a time)")
        out["roofline"] = roof
        if world == 1 and not args.no_extras and not args.no_other_workloads and args.workload == DEFAULT_WORKLOAD:
            others = {}
            for name, n_frames in OTHER_WORKLOADS.items():
                others[name] = run_other_workload(name, n_frames, rt, Scene, torch, cpu=not args.no_cpu_baseline)
            out["other_workloads"] = others
        if others_multi is not None:
            out["other_workloads"] = others_multi
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
