#!/usr/bin/env python3
"""Benchmark of the ray-scene query path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One step = one frame of the workload rendered through the C ABI (rt_render): primary rays, closest
hits, area-light shadow rays (any-hit), reflection / refraction levels, with the scene resident in
HBM.  For N > 1 (launched with torch.distributed.run, one rank per GPU) the image is cut into bands
of 8 pixel rows dealt round-robin to the ranks, the scene is replicated, and the bands are gathered
to rank 0 over RCCL inside the timed region (the gather of a frame overlaps the rendering of the next
one; all of them are complete before the clock stops); the total work is fixed ("strong" scaling).

Prints ONE JSON line on rank 0.  metric = Mrays/s, one ray = one Search-equivalent query of the
reference (primary, secondary or shadow).  `roofline` prices the dominant kernel (k_shadow: the
any-hit pass over all area-light shadow rays) against the FP32 vector peak using the REFERENCE algorithm's work per ray
(3 + 18*B + 52*T flops; B slab tests and T ray/triangle tests counted by the CPU restatement), and
`cpu_baseline` is that restatement timed on one host core on a bounded sample of the same frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
        shading="flat", cull=True, max_depth=10, aspect="true", cpu_rows=(1000, 1160)),
    # fuzzysampling (64 samples per pixel) + 32 samples per light, 4K
    "cornell_original_blinn_spp64_spt32_4k": dict(
        scene="cornell_original", cam=CAM_ORIGINAL, width=3840, height=2160, spp=64, spt=32, render_type="blinn_phong",
        shading="flat", cull=True, max_depth=10, cpu_rows=(1080, 1082)),
    # a small case for quick checks
    "cornell_original_blinn_spt4_256": dict(
        scene="cornell_original", cam=CAM_ORIGINAL, width=256, height=256, spp=1, spt=4, render_type="blinn_phong",
        shading="flat", cull=True, max_depth=10, cpu_rows=(96, 160)),
}
DEFAULT_WORKLOAD = "cornell_original_blinn_spt16_1080p"

FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X: 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2.4 GHz
FP32_NO_FMA_PEAK_TOPS = 78.6      # the same lanes issuing one non-fused op per clock (bit-parity forbids FMA)
HBM_PEAK_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
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
    if wl["scene"].startswith("synthetic:"):
        from raytracer_amd.synthetic import cornell_with_icosphere
        box = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", "cornell_original.npz"))
        scene = cornell_with_icosphere(box, int(wl["scene"].split(":")[1]))
    else:
        scene = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scenes", wl["scene"] + ".npz"))
    ads = rt.GpuOctree({"max_triangles_per_BB": 16, "max_depth": 10}, device=local_rank)
    ads.Build(scene.pos)
    ads.set_shading(scene.norm, scene.matid, scene.materials)
    tree_stats = ads.tree_stats()
    W, H = wl["width"], wl["height"]
    aspect = 1 if wl.get("aspect") == "true" else 0
    kw = dict(fov=0.6, width=W, height=H, spp=wl["spp"], spt=wl["spt"], render_type=wl["render_type"], shading=wl["shading"],
              cull=wl["cull"], max_depth=wl["max_depth"], seed=42, aspect_mode=aspect, **wl["cam"])
    cfg = rt.make_render_config(band_rank=rank, band_world=world, **kw)

    from raytracer_amd import distributed as rd
    # Two frame buffers: while the bands of one frame travel to rank 0 (RCCL, on its own stream), the next frame is
    # rendered into the other buffer.  RT_BENCH_SYNC_GATHER=1 completes every gather before the next frame starts.
    frames = [rd.alloc_frame(H, W, world, "cuda") for _ in range(2)]
    frame = frames[0]
    stream = torch.cuda.current_stream().cuda_stream
    gathers = [rd.BandGather(H, W, rank, world, "cuda", dist) for _ in range(2)]
    overlap = dist is not None and os.environ.get("RT_BENCH_SYNC_GATHER") != "1"
    step_no = [0]

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        gathers[k].finish()  # this buffer's previous frame must have arrived before it is overwritten
        _, st = ads.render(cfg, out_ptr=frames[k].data_ptr(), stream=stream)
        gathers[k].start(frames[k])  # framebuffer bands -> rank 0; no-op for one rank
        if not overlap:
            gathers[k].finish()
        return st

    def fence():
        for g_ in gathers:
            g_.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0
    shade_ms = closest_ms = shadow_ms = 0.0
    shade_launches = closest_launches = shadow_launches = 0
    shadow_rays = closest_rays = 0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
        shade_ms += st.ms_shade
        closest_ms += st.ms_closest
        # per-kernel times come from HIP events around the first 64 launches of each class in a call; the ray
        # counts used for kernel rates are those of exactly these launches
        shade_launches += min(st.launches_shade, 64)
        shadow_ms += st.ms_shadow
        shadow_launches += min(st.launches_shadow, 64)
        closest_launches += min(st.launches_closest, 64)
        shadow_rays += st.timed_shadow_rays
        closest_rays += st.timed_closest_rays
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays], dtype=torch.int64, device="cuda")
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays = int(rr.item())

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
                       "triangles": scene.n_triangles, "octree_nodes": tree_stats["nodes"], "octree_build_s": round(tree_stats["build_seconds"], 3),
                       "rays_per_frame": rays // args.steps, "partition": f"8-row bands round-robin over {world} rank(s)"},
        }
        # ---- CPU baseline + reference-work counts on a bounded sample of the same frame (rank 0, N = 1 only) ----
        flops_any = flops_closest = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc
            osc = orc.Scene.from_arrays(scene.pos, scene.norm, scene.matid, scene.materials)
            otree = orc.Octree(osc, 16, 10)
            r0, r1 = wl["cpu_rows"]
            ocfg = orc.make_cfg(rng_mode=orc.RNG_COUNTER, rows=(r0, r1), **kw)
            c0 = time.perf_counter()
            ref, _, _, ost = orc.render(osc, otree, ocfg, want_hits=False)
            cpu_s = time.perf_counter() - c0
            img = frame[:H].cpu().numpy()
            diff = float(np.abs(img[r0:r1] - ref[r0:r1]).max())
            out["cpu_baseline"] = {"value": ost.search_calls / cpu_s / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
                                   "sample": f"rows {r0}..{r1 - 1} of the same frame ({ost.search_calls} rays, {cpu_s:.1f} s), "
                                             "oracle/rt_oracle.c -O2, counter RNG"}
            out["parity_on_sample"] = {"max_abs_diff": diff, "tolerance": 1.0 / 255.0, "ok": diff <= 1.0 / 255.0}
            n_any = max(ost.shadow_rays, 1)
            n_closest = max(ost.primary_rays + ost.secondary_rays, 1)
            b_any, t_any = ost.slab_tests_any / n_any, ost.tri_tests_any / n_any
            b_cl = (ost.slab_tests - ost.slab_tests_any) / n_closest
            t_cl = (ost.tri_tests - ost.tri_tests_any) / n_closest
            flops_any = 3 + 18 * b_any + 52 * t_any
            flops_closest = 3 + 18 * b_cl + 52 * t_cl
            out["reference_work_per_ray"] = {"any_hit": {"B": b_any, "T": t_any, "flops": flops_any},
                                             "closest_hit": {"B": b_cl, "T": t_cl, "flops": flops_closest}}
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
                    "value": rr["search_calls"] / rr["wall_s"] / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                    "sample": f"oracle/_ref/raytracer_ref (-O2, glibc rand) on the same scene and settings at {rw}x{rh}, 1 sample "
                              f"per pixel: {rr['search_calls']} Search calls in {rr['wall_s']:.1f} s (whole process)"}
        if world == 1:
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
        kernel_rate = shadow_rays / max(shadow_ms * 1e-3, 1e-12) / 1e6
        roof = {"bound": "valu", "kernel": "k_shadow (any-hit pass: area-light shadow rays)", "achieved": None,
                "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                "launches": shadow_launches, "avg_launch_ms": shadow_ms / max(shadow_launches, 1),
                "rays_per_launch": shadow_rays / max(shadow_launches, 1), "kernel_Mrays_per_s": kernel_rate,
                "other_kernels": {"k_closest": {"launches": closest_launches, "avg_launch_ms": closest_ms / max(closest_launches, 1),
                                                "Mrays_per_s": closest_rays / max(closest_ms * 1e-3, 1e-12) / 1e6},
                                  "k_shade": {"launches": shade_launches, "avg_launch_ms": shade_ms / max(shade_launches, 1)}}}
        if flops_any is not None and shadow_ms > 0:
            tflops = shadow_rays * flops_any / (shadow_ms * 1e-3) / 1e12
            roof.update({"achieved": tflops, "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
                         "frac_of_no_fma_issue_peak": tflops / FP32_NO_FMA_PEAK_TOPS, "algorithmic_flops_per_ray": flops_any})
            roof["other_kernels"]["k_closest"]["achieved_TFLOPs"] = closest_rays * flops_closest / max(closest_ms * 1e-3, 1e-12) / 1e12
        # HBM bytes per k_shadow launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this
        # same command; they cannot be collected from inside the run, so the committed summary is quoted.
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        if pmc_files and world == 1:
            with open(pmc_files[-1]) as f:
                pmc = json.load(f)
            if pmc.get("workload") == args.workload:
                roof["traffic"] = pmc["k_shadow_bytes_per_launch_corrected"]["mean_over_launches"]
                roof["traffic_source"] = (f"profiles/{os.path.basename(pmc_files[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                          "separate runs, reads doubled per the gfx950 rule; bytes per launch)")
                # per hit: work index + t + triangle index read (12 B), one visibility word per light and 32 samples written
                n_lights = max(len(scene.light_triangles()), 1)
                hits = (shadow_rays / max(shadow_launches, 1)) / (wl["spt"] * n_lights)
                roof["algorithmic_bytes_per_launch"] = hits * (12.0 + 4.0 * n_lights * ((wl["spt"] + 31) // 32))
        out["roofline"] = roof
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
