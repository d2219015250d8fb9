#!/usr/bin/env python3
"""Benchmark of the render loop on MI355X.

A step is one full frame of the headline workload (BASELINE.json: "dragon.obj 1080p 16spp soft shadows"):
dragon-class mesh (data/dragon.obj is absent from the reference checkout, so a fixed-seed 81,920-triangle
stand-in is synthesised, see ray-tracer_amd/synth.py), 1920x1080, the reference's default camera, Phong
shading + parallelogram-light soft shadows (4 x 4 = 16 light samples per hit), 16 samples per pixel
(ray_multiplier 4), BVH on.  Scene, hierarchy and textures are resident in HBM before the timed region; the
frame is rendered into device memory.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is split into interleaved blocks of
16 rows, every rank renders its blocks from its own replica of the scene, and the blocks are gathered on
rank 0 over RCCL inside the timed region.  Total work is fixed, so scaling is "strong".

Rank 0 prints one JSON line.  `value` is rays traced per second over the whole job (a ray = one closest-hit
or any-hit query against the hierarchy: primary, shadow, bounce), in Mrays/s.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
FP32_PEAK_TFLOPS = 157.3   # MI355X fp32 vector peak (packed fused multiply-add in every lane)
FP32_NO_FMA_TFLOPS = 78.6  # the same without fused multiply-add: this path is compiled -ffp-contract=off by contract
# The compiled reference as it is against the oracle, one thread each, on THIS benchmark's mesh (81,920 triangles, same
# light, shading + soft shadows, 16 spp, 48x27 pixels = 75,200 rays): 101.3 s against 0.340 s, measured in the build
# container (8-vCPU Xeon VM, g++ 11.4) by tools/time_reference_headline.py.
REFERENCE_AS_IS_RATIO = 298.0
L2_PEAK_GBS = 34500.0      # MI355X aggregate L2 bandwidth (8 XCDs)
CPU_ROW_STEP = 12          # the CPU leg renders every 12th row of the frame
# the kernels of one frame of the headline workload as rocprofv3 names them (timed variants: test counting off)
FRAME_KERNELS = ["k_primary<true, false>", "k_stream<false, true, 1>", "k_closest<true, false>", "k_cull", "k_stream<false, false, 2>", "k_shade", "k_resolve"]
CAMERA = dict(fov_deg=50.0, distance=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0))  # reference src/config.h:16-21


def algorithmic_bytes(stats, pixels):
    """28 B ray in + 16 B hit out per query, 32 B per ray/box test (24 B box + 8 B child reference), 36 B per
    ray/triangle test (nine floats), 12 B per pixel written."""
    rays = stats.primary_rays + stats.shadow_rays + stats.bounce_rays + stats.probe_rays
    return 44 * rays + 32 * stats.box_tests + 36 * stats.triangle_tests + 12 * pixels


def algorithmic_flops(stats):
    """27 per ray/box test, 150 per ray/triangle test (sqrt and divide counted as one each), 60 + one pow per
    shaded light sample."""
    return 27 * stats.box_tests + 150 * stats.triangle_tests + 61 * stats.shadow_rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mesh-level", type=int, default=6, help="icosphere subdivisions: 6 = 81,920 triangles, 8 = 1,310,720")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (the roofline numerator then falls back to the kernel's own test counts)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra 1.31 M-triangle measurement")
    ap.add_argument("--workload", choices=["c3", "c5"], default="c3",
                    help="c3 (default): the headline workload, BASELINE.json configs[2]; c5: configs[4] in the form the reference can express "
                         "(3840x2160, soft shadows + depth of field with 63 lens rays + bloom, no supersampling)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the last assembled frame to this .npy file")
    args = ap.parse_args()
    global WIDTH, HEIGHT, CPU_ROW_STEP
    if args.workload == "c5":
        WIDTH, HEIGHT, CPU_ROW_STEP = 3840, 2160, 96

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a box with a single GPU: every rank uses GPU 0 and the exchange runs on gloo
    rehearsal = os.environ.get("BENCH_ONE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    rt = importlib.import_module("ray-tracer_amd")
    synth = importlib.import_module("ray-tracer_amd.synth")
    tiling = importlib.import_module("ray-tracer_amd.tiling")
    lib = rt.Library()  # raises when librt_amd.so is missing: there is no fallback path
    lib.init(local_rank)

    mesh = synth.dragon_class_mesh(args.mesh_level)
    scene = rt.SceneData([mesh], lights=synth.area_light())
    sh = lib.scene_create(scene)
    bvh = lib.bvh_build(sh)
    lights, n_lights = scene.light_array()
    cam = rt.make_camera(width=WIDTH, height=HEIGHT, **CAMERA)
    if args.workload == "c5":
        feats = rt.features(shading=1, soft_shadow=1, depth_of_field=1, bloom=1)
        param_args = dict(dof_samples=63, dof_f=2.5, dof=0.05, dof_aperture=1.4)
        label = "3840x2160 soft shadows (16 light samples) + depth of field (63 lens rays) + bloom"
    else:
        feats = rt.features(shading=1, soft_shadow=1, multiple_rays=1)
        param_args = dict(ray_multiplier=4)
        label = "1080p 16spp soft shadows (16 light samples)"
    params = rt.default_params(**param_args)

    # interleaved 16-row blocks, block b belongs to rank b % world (ray-tracer_amd/tiling.py)
    my_blocks = tiling.blocks_of_rank(rank, world, HEIGHT)
    local = tiling.local_buffer(world, WIDTH, HEIGHT, dev)
    gathered = tiling.gather_buffers(local, world) if (world > 1 and rank == 0) else None
    row_index = tiling.frame_row_index(world, HEIGHT, dev) if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream
    # BENCH_GATHER=native: the blocks travel by the library's own grouped RCCL sends / receives (rt_gather_blocks)
    # instead of torch.distributed's gather; needs a GPU per rank (RCCL refuses two ranks on one device)
    native = world > 1 and os.environ.get("BENCH_GATHER") == "native" and not os.environ.get("BENCH_ONE_GPU_REHEARSAL")
    comm = tiling.native_comm(lib, rank, world) if native else None
    native_frame = torch.empty((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev) if (native and rank == 0) else None

    def render_frame(p):
        """Renders this rank's rows into `local`; returns the summed rt_stats of its rt_render calls."""
        total = rt.Stats()

        def render_rows(row_begin, row_end, out):
            _, st = lib.render(bvh, cam, lights, n_lights, feats, p, WIDTH, HEIGHT, row_begin, row_end, out=out.data_ptr(), device_ptr=1, stream=stream)
            for name, _ in rt.Stats._fields_:
                setattr(total, name, getattr(total, name) + getattr(st, name))

        if world == 1:
            render_rows(0, HEIGHT, local)
        else:
            # this rank's interleaved row blocks in one call, laid out like tiling.local_buffer
            _, st = lib.render_blocks(bvh, cam, lights, n_lights, feats, p, WIDTH, HEIGHT, tiling.ROW_BLOCK, rank, world,
                                      out=local.data_ptr(), device_ptr=1, stream=stream)
            for name, _ in rt.Stats._fields_:
                setattr(total, name, getattr(total, name) + getattr(st, name))
        return total

    last = {}

    def step():
        st = render_frame(params)
        if native:
            frame = tiling.gather_frame_native(lib, comm, local, rank, world, WIDTH, HEIGHT, frame=native_frame, stream=stream)
        else:
            frame = tiling.gather_frame(local, rank, world, WIDTH, HEIGHT, gathered=gathered, row_index=row_index)
        if rank == 0:
            if feats.bloom:  # the whole-frame post pass runs where the frame is assembled (reference src/render.cpp:125-127)
                frame = frame.contiguous()
                lib.bloom_device(frame.data_ptr(), WIDTH, HEIGHT, params, stream)
            last["frame"] = frame
        return st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, shadow_ms, shadow_b_ms, primary_ms, shade_ms, cull_ms, rays, traced = 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0
    for _ in range(args.steps):
        st = step()
        kernel_ms += st.kernel_ms
        shadow_ms += st.shadow_ms
        shadow_b_ms += st.shadow_b_ms
        primary_ms += st.primary_ms
        shade_ms += st.shade_ms
        cull_ms += st.cull_ms
        rays += st.reference_rays()
        traced += st.total_rays()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, float(rays), float(traced)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, rays, traced = float(tmax[0]), int(tsum[1]), int(tsum[2])

    # one untimed frame with test counting on: the tests the kernels themselves perform (counts are deterministic)
    count_params = rt.default_params(count_tests=1, **param_args)
    cst = render_frame(count_params)
    torch.cuda.synchronize()
    my_pixels = WIDTH * sum(tiling.block_rows(b, HEIGHT)[1] - tiling.block_rows(b, HEIGHT)[0] for b in my_blocks)
    have_split = shadow_ms > 0.0
    avg_kernel_s = kernel_ms / args.steps / 1e3
    # The shadow queries of a frame are answered by two launches: the main round (k_stream<false, false, 2>) and, on the
    # library's second stream beside it, the small round for the hits of the camera rays that were finished late
    # (k_stream<false, false, 1>).  The test counts below cover both, so the time does too: the SUM of the two launches'
    # durations, although they overlap (the second one's includes its wait for wave slots): a conservative denominator.
    dom_s = ((shadow_ms + shadow_b_ms) if have_split else kernel_ms) / args.steps / 1e3

    if rank == 0:
        # CPU leg (rank 0 only): the oracle on every 12th row of the same frame.  It is the CPU baseline, and its
        # counters are the ALGORITHM's ray/box and ray/triangle tests in the reference's traversal order (every shadow
        # query traced as a closest-hit query, none skipped), which is what the roofline numerator is made of.
        cpu = None if args.no_cpu_baseline else cpu_baseline(rt, scene, cam, feats, params)
        share = my_pixels / float(WIDTH * HEIGHT)  # rank 0's part of the frame (interleaved blocks: a fair sample)
        if cpu is not None:
            ost = cpu.pop("_stats")
            lit = cpu.pop("_lit")
            scale = cpu.pop("_scale") * share
            # the shadow pass processes the queries whose light sample the surface faces (the others are answered by
            # the cull pass): its numerator is the oracle's tests on exactly those queries, in the reference's order
            ref_counts = {"source": "oracle (reference traversal order, each query a closest-hit query as in the reference), every "
                                    f"{CPU_ROW_STEP}th row scaled to " + ("the frame" if world == 1 else "rank 0's share of the frame"),
                          "shadow_rays": lit[0] * scale, "shadow_box_tests": lit[1] * scale, "shadow_triangle_tests": lit[2] * scale,
                          "all_shadow_rays": ost.shadow_rays * scale, "all_shadow_box_tests": ost.shadow_box_tests * scale,
                          "all_shadow_triangle_tests": ost.shadow_triangle_tests * scale, "box_tests": ost.box_tests * scale,
                          "triangle_tests": ost.triangle_tests * scale, "rays": ost.total_rays() * scale}
        else:
            ref_counts = {"source": "the kernels' own test counts (CPU leg skipped)", "shadow_rays": cst.shadow_rays,
                          "shadow_box_tests": cst.shadow_box_tests, "shadow_triangle_tests": cst.shadow_triangle_tests,
                          "box_tests": cst.box_tests, "triangle_tests": cst.triangle_tests, "rays": cst.total_rays()}
        if have_split:
            dom_name = ("k_stream<false, false, 2> (shadow pass) + k_stream<false, false, 1> (the second, small lighting round on the "
                        "library's own stream): kernel_ms is the sum of the two launches' durations, the test counts cover both")
            dom_key = "k_stream<false, false, 2>"
            alg_flops = 27 * ref_counts["shadow_box_tests"] + 150 * ref_counts["shadow_triangle_tests"]
            alg_bytes = 44 * ref_counts["shadow_rays"] + 32 * ref_counts["shadow_box_tests"] + 36 * ref_counts["shadow_triangle_tests"]
            own_flops = 27 * cst.shadow_box_tests + 150 * cst.shadow_triangle_tests
            own_bytes = 44 * cst.shadow_rays + 32 * cst.shadow_box_tests + 36 * cst.shadow_triangle_tests
        else:
            dom_name, dom_key = "all kernels of rt_render", None
            alg_flops = 27 * ref_counts["box_tests"] + 150 * ref_counts["triangle_tests"]
            alg_bytes = 44 * ref_counts["rays"] + 32 * ref_counts["box_tests"] + 36 * ref_counts["triangle_tests"] + 12 * my_pixels
            own_flops, own_bytes = algorithmic_flops(cst), algorithmic_bytes(cst, my_pixels)
        counters = measured_counters(world)
        dom_counters = (counters or {}).get("per_kernel", {}).get(dom_key, {}) if dom_key else {}
        traffic = dom_counters.get("hbm_bytes_per_launch")
        l2_req = dom_counters.get("l2_read_requests_per_launch")
        frame_ms = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s, dragon-class mesh " + label,
            "value": traced / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": frame_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[2]: dragon-class mesh ({scene.n_triangles} triangles, synthetic stand-in for the absent dragon.obj), "
                             f"{WIDTH}x{HEIGHT}, BVH on, Phong + parallelogram light soft shadows (16 samples), 16 spp") if args.workload == "c3" else
                            (f"configs[4] as the reference can express it: dragon-class mesh ({scene.n_triangles} triangles), {WIDTH}x{HEIGHT}, BVH on, Phong + "
                             "soft shadows (16 samples) + depth of field (centre ray + 63 lens rays where the pixel is out of focus) + bloom on the "
                             "assembled frame; no supersampling (the reference skips depth of field when it is on), no glossy reflection"),
                "triangles": scene.n_triangles,
                "bvh_levels": int(lib.lib.rt_bvh_num_levels(bvh)),
                "rays_per_frame": traced // args.steps,
                "reference_rays_per_frame": rays // args.steps,
                "ray_accounting": "value counts the queries the kernels trace (rays_per_frame).  The reference's algorithm issues "
                                  "reference_rays_per_frame for the same frame: shadow queries whose Phong factor is exactly zero "
                                  "(surface facing away from the light sample) are answered without a ray here, the image is the same.  "
                                  "The CPU baseline traces all of them: compare it with reference_workload_mrays_per_s",
                "partition": "whole frame" if world == 1 else f"interleaved {tiling.ROW_BLOCK}-row blocks over {world} ranks, gather to rank 0",
            },
            "frame_ms": frame_ms,
            "reference_workload_mrays_per_s": rays / elapsed / 1e6,
            # The dominant kernel against the roofs that can bind it.  Its working set (hierarchy + primitives, about
            # 10 MB) stays in the caches, so HBM is not its roof (hbm.frac below, from the measured traffic); its
            # vector issue slots are 86 % in use (decomposition.valu_issue_util), so the headline fraction is fp32
            # VALU.  No fraction uses a numerator other than what its field says.
            "roofline": {
                "kernel": dom_name,
                "bound": "valu",
                # headline fraction: the tests the launches themselves perform (counted by the kernels) over the fp32 vector peak
                "achieved": own_flops / dom_s / 1e12,
                "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": own_flops / dom_s / 1e12 / FP32_PEAK_TFLOPS,
                "traffic": traffic,
                "kernel_ms": dom_s * 1e3,
                "numerator": "27 flops per ray/box test + 150 per ray/triangle test (sqrt and divide counted as one each) over the tests "
                             "the shadow launches perform (any-hit: a query stops at its first accepted triangle), counted by the kernels "
                             "in an untimed frame of the same workload",
                "queries_per_launch": {"oracle": ref_counts["shadow_rays"], "kernel": cst.shadow_rays} if have_split else None,
                "algorithmic_flops_per_launch": own_flops,
                "algorithmic_bytes_per_launch": own_bytes,
                # the same against the peak this code can reach at all: no fused multiply-add by contract (bit-exact decisions)
                "frac_of_no_fma_peak": own_flops / dom_s / 1e12 / FP32_NO_FMA_TFLOPS,
                "no_fma_peak": FP32_NO_FMA_TFLOPS,
                # what the fraction is made of (counters of the committed profile of this command, see counters_from)
                "decomposition": None if not dom_counters.get("valu_issue_util") else {
                    "valu_issue_util": round(dom_counters["valu_issue_util"], 3),
                    "active_lane_frac": round(dom_counters["active_lane_frac"], 3),
                    "useful_flops_per_issued_lane_op": round(own_flops / (dom_counters["valu_insts_per_launch"] * 64.0 * dom_counters["active_lane_frac"]), 3),
                    "lane_ops_per_box_test": round(dom_counters["valu_insts_per_launch"] * 64.0 / max(1, cst.shadow_box_tests), 1),
                    "note": "valu_issue_util = SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel cycles): share of the chip's vector issue slots in use; "
                            "active_lane_frac = lanes enabled per issued instruction; useful flops per issued lane operation = algorithmic flops "
                            "/ (vector instructions x 64 x active lanes): the rest is addressing, compaction, votes and the stream's bookkeeping; "
                            "the peak counts 2 flops per lane operation (fused multiply-add) and 2 lanes' worth per packed instruction, neither "
                            "of which non-contracted scalar code can use (only the slab subtract/multiply of the box test is packed)"},
                "reference_order": None if "all_shadow_box_tests" not in ref_counts or not have_split else {
                    "note": "the same with the oracle's counts for the queries these launches answer, each traced as the reference does (a "
                            "closest-hit query in its traversal order) - the algorithm's own work rather than this implementation's; and over "
                            "ALL shadow queries of the reference's algorithm, including those the cull pass answers without a ray (for scale only)",
                    "flops_per_launch": alg_flops, "bytes_per_launch": alg_bytes,
                    "achieved": alg_flops / dom_s / 1e12, "frac": alg_flops / dom_s / 1e12 / FP32_PEAK_TFLOPS,
                    "all_queries_flops": 27 * ref_counts["all_shadow_box_tests"] + 150 * ref_counts["all_shadow_triangle_tests"],
                    "counts_from": ref_counts["source"]},
                "hbm": None if traffic is None else {"achieved": traffic / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": traffic / dom_s / 1e9 / HBM_PEAK_GBS, "source": "FETCH_SIZE / WRITE_SIZE passes, profiles/traffic.json"},
                "l2": None if l2_req is None else {"read_requests_per_launch": l2_req, "achieved_at_64B": l2_req * 64 / dom_s / 1e9,
                                                   "achieved_at_128B": l2_req * 128 / dom_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                                                   "frac_at_64B": l2_req * 64 / dom_s / 1e9 / L2_PEAK_GBS,
                                                   "frac_at_128B": l2_req * 128 / dom_s / 1e9 / L2_PEAK_GBS,
                                                   "source": "TCP_TCC_READ_REQ_sum, profiles/traffic.json"},
                "active_lane_frac": None if counters is None else {k: round(counters["per_kernel"][k]["active_lane_frac"], 3) for k in FRAME_KERNELS
                                                                   if "active_lane_frac" in counters["per_kernel"].get(k, {})},
                "counters_from": None if counters is None else counters.get("source"),
                "all_kernels": {"kernel_ms": avg_kernel_s * 1e3, "primary_ms": primary_ms / args.steps, "shadow_ms": shadow_ms / args.steps,
                                "shadow_second_round_ms": shadow_b_ms / args.steps,
                                "cull_ms": cull_ms / args.steps, "shade_ms": shade_ms / args.steps,
                                "own_tests_bytes": algorithmic_bytes(cst, my_pixels), "own_tests_flops": algorithmic_flops(cst)},
                "stream_steps": {"packets": cst.stream_packets, "box_steps": cst.stream_box_steps, "primitive_steps": cst.stream_prim_steps,
                                 "entries_per_box_step": cst.stream_entries / max(1, cst.stream_box_steps),
                                 "note": "shadow pass of the counted frame: packets of 64 queries, steps of the pair stream"},
                "box_tests_per_ray": cst.box_tests / max(1, cst.total_rays()),
                "triangle_tests_per_ray": cst.triangle_tests / max(1, cst.total_rays()),
            },
        }
        if cpu is not None and world == 1:
            out["cpu_baseline"] = cpu
        if args.dump_frame:
            torch.cuda.synchronize()
            np.save(args.dump_frame, last["frame"].cpu().numpy())
        if world == 1 and not args.no_extra and args.workload == "c3":
            out["extra"] = {"headline_with_sah_hierarchy": extra_sah_hierarchy(rt, lib, sh, scene, cam, feats, params, local, stream),
                            "mesh_level_8": extra_large_mesh(rt, synth, lib, cam, feats, params, local, stream),
                            "teapot": extra_teapot(rt, lib, cam, feats, params, local, stream)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_counters(world):
    """Per-kernel hardware counters from the committed rocprofv3 PMC passes of this same command at one GPU
    (profiles/traffic.json, written by profiles/make_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate runs,
    in KiB, FETCH_SIZE doubled as prescribed for gfx950; L2 read requests; active-lane fraction), or None when no
    measurement is committed.  They describe a whole-frame launch, so they are only attached at one GPU."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def cpu_baseline(rt, scene, cam, feats, params):
    """The oracle (oracle/cpu_ref.cpp, the CPU restatement of the reference's algorithm) timed on this
    machine's host cores on every 12th row of the same frame at full sample count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    osc = o.OracleScene(scene)
    t0 = time.perf_counter()
    _, st = osc.render(cam, feats, params, WIDTH, HEIGHT, row_step=CPU_ROW_STEP, threads=threads)
    dt = time.perf_counter() - t0
    lit = o.last_lit_shadow_counts()
    rows = (HEIGHT + CPU_ROW_STEP - 1) // CPU_ROW_STEP
    return {
        "value": st.total_rays() / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"every {CPU_ROW_STEP}th row ({rows} rows) of the same {WIDTH}x{HEIGHT} frame at full sample count ({st.total_rays()} rays, {dt:.1f} s, OpenMP schedule(guided) over rows)",
        "frame_ms_extrapolated": dt * HEIGHT / rows * 1e3,
        "compare_with": "reference_workload_mrays_per_s (the CPU leg traces every shadow query of the reference's algorithm)",
        "reference_as_is_ratio": REFERENCE_AS_IS_RATIO,
        "reference_as_is_note": "the compiled reference as it is takes this many times the oracle's time on this benchmark's mesh (one thread "
                                "each, 48x27 pixels at 16 spp, build container, tools/time_reference_headline.py): it copies the whole Mesh "
                                "per leaf primitive it tests (src/bounding_volume_hierarchy.cpp:339); value / reference_as_is_ratio is the "
                                "reference's own rate per thread-equivalent",
        "_stats": st,
        "_lit": lit,
        "_scale": HEIGHT / float(rows),
    }


def extra_teapot(rt, lib, cam, feats, params, out, stream):
    """The same frame on the largest real mesh of the reference's data directory (teapot.obj, 15,704 triangles, with the
    reference's parallelogram-light scene around it), from the scene arrays of the golden fixtures."""
    path = os.path.join(ROOT, "tests", "golden", "teapot.npz")
    if not os.path.exists(path):
        return None
    light = np.load(os.path.join(ROOT, "tests", "golden", "cornell_para.npz"))["scene.lights"]
    g = np.load(path)
    scene = rt.SceneData.from_golden(g, skybox=False)
    bvh = lib.bvh_build(lib.scene_create(scene))
    lights, n_lights = scene.light_array(light)
    best, st = None, None
    for _ in range(4):
        _, st = lib.render(bvh, cam, lights, n_lights, feats, params, WIDTH, HEIGHT, 0, HEIGHT, out=out.data_ptr(), device_ptr=1, stream=stream)
        best = st.kernel_ms + st.resolve_ms if best is None else min(best, st.kernel_ms + st.resolve_ms)
    return {"triangles": scene.n_triangles, "bvh_levels": int(lib.lib.rt_bvh_num_levels(bvh)), "frame_ms": best,
            "mrays_per_s": st.total_rays() / best / 1e3, "reference_workload_mrays_per_s": st.reference_rays() / best / 1e3,
            "note": "teapot.obj under the reference's parallelogram light (CornellBoxParallelogramLight's), device time of rt_render, best of 4"}


def extra_sah_hierarchy(rt, lib, sh, scene, cam, feats, params, out, stream):
    """The headline frame over the hierarchy the reference builds with its enableBvhSahBinning toggle (64 bins) instead
    of the default median split: same pixels' worth of work, a tree with tighter boxes."""
    lib.lib.rt_bvh_destroy(lib.bvh_build_device(sh, sah=True)[0])  # (first call: module load, allocator warm-up)
    t0 = time.perf_counter()
    bvh, on_device = lib.bvh_build_device(sh, sah=True)
    build_s = time.perf_counter() - t0
    lights, n_lights = scene.light_array()
    best, st = None, None
    for _ in range(6):
        _, st = lib.render(bvh, cam, lights, n_lights, feats, params, WIDTH, HEIGHT, 0, HEIGHT, out=out.data_ptr(), device_ptr=1, stream=stream)
        best = st.kernel_ms + st.resolve_ms if best is None else min(best, st.kernel_ms + st.resolve_ms)
    res = {"triangles": scene.n_triangles, "bvh_levels": int(lib.lib.rt_bvh_num_levels(bvh)), "device_build_s": build_s, "built_on_device": bool(on_device),
           "frame_ms": best, "mrays_per_s": st.total_rays() / best / 1e3, "primary_ms": st.primary_ms, "cull_ms": st.cull_ms,
           "shadow_ms": st.shadow_ms, "shade_ms": st.shade_ms, "note": "device time of rt_render (HIP events), best of 6"}
    lib.lib.rt_bvh_destroy(bvh)
    return res


def extra_large_mesh(rt, synth, lib, cam, feats, params, out, stream):
    """The same frame on the 1,310,720-triangle mesh (hierarchy and primitives no longer fit the L2 caches)."""
    mesh = synth.dragon_class_mesh(8)
    scene = rt.SceneData([mesh], lights=synth.area_light())
    sh = lib.scene_create(scene)
    t0 = time.perf_counter()
    host_bvh = lib.bvh_build(sh)
    build_s = time.perf_counter() - t0
    lib.lib.rt_bvh_destroy(host_bvh)
    import torch

    lib.lib.rt_bvh_destroy(lib.bvh_build_device(sh)[0])  # (first call: module load, allocator warm-up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bvh, on_device = lib.bvh_build_device(sh)
    device_build_s = time.perf_counter() - t0
    lights, n_lights = scene.light_array()
    best, st = None, None
    for _ in range(4):
        _, st = lib.render(bvh, cam, lights, n_lights, feats, params, WIDTH, HEIGHT, 0, HEIGHT, out=out.data_ptr(), device_ptr=1, stream=stream)
        best = st.kernel_ms + st.resolve_ms if best is None else min(best, st.kernel_ms + st.resolve_ms)
    out_d = {"triangles": scene.n_triangles, "bvh_levels": int(lib.lib.rt_bvh_num_levels(bvh)), "host_build_s": build_s,
             "device_build_s": device_build_s, "built_on_device": bool(on_device),
             "frame_ms": best, "mrays_per_s": st.total_rays() / best / 1e3, "reference_workload_mrays_per_s": st.reference_rays() / best / 1e3,
             "primary_ms": st.primary_ms, "cull_ms": st.cull_ms, "shadow_ms": st.shadow_ms, "shade_ms": st.shade_ms,
             "note": "device time of rt_render (HIP events), best of 4"}
    # the shadow pass of this frame against the memory roofs, from the committed counters of the same frame (131 MB of
    # hierarchy and primitives: beyond the L2 caches)
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_l8.json")) as f:
            k = json.load(f)["per_kernel"]["k_stream<false, false, 2>"]
        sec = st.shadow_ms / 1e3
        out_d["shadow_pass"] = {
            "hbm": {"bytes_per_launch": k["hbm_bytes_per_launch"], "achieved": k["hbm_bytes_per_launch"] / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": k["hbm_bytes_per_launch"] / sec / 1e9 / HBM_PEAK_GBS,
                    "note": "FETCH_SIZE / WRITE_SIZE count what leaves the L2 caches; part of it is served by the 256 MB Infinity Cache, so this is an upper bound of the HBM share"},
            "l2": {"read_requests_per_launch": k["l2_read_requests_per_launch"], "achieved_at_128B": k["l2_read_requests_per_launch"] * 128 / sec / 1e9,
                   "peak": L2_PEAK_GBS, "unit": "GB/s", "frac_at_64B": k["l2_read_requests_per_launch"] * 64 / sec / 1e9 / L2_PEAK_GBS,
                   "frac_at_128B": k["l2_read_requests_per_launch"] * 128 / sec / 1e9 / L2_PEAK_GBS},
            "valu_issue_util": round(k["valu_issue_util"], 3), "active_lane_frac": round(k["active_lane_frac"], 3),
            "counters_from": "profiles/r3_l8/summary.txt (profiles/traffic_l8.json)"}
    except (OSError, ValueError, KeyError):
        pass
    lib.lib.rt_bvh_destroy(bvh)
    # the surface-area-heuristic variant of the hierarchy (four sorts per node, 64 bins): build times and the same frame
    t0 = time.perf_counter()
    host_bvh = lib.bvh_build(sh, sah=True)
    sah_host_s = time.perf_counter() - t0
    lib.lib.rt_bvh_destroy(host_bvh)
    t0 = time.perf_counter()
    bvh, on_device = lib.bvh_build_device(sh, sah=True)
    sah_device_s = time.perf_counter() - t0
    best = None
    for _ in range(4):
        _, st = lib.render(bvh, cam, lights, n_lights, feats, params, WIDTH, HEIGHT, 0, HEIGHT, out=out.data_ptr(), device_ptr=1, stream=stream)
        best = st.kernel_ms + st.resolve_ms if best is None else min(best, st.kernel_ms + st.resolve_ms)
    out_d["sah_64_bins"] = {"bvh_levels": int(lib.lib.rt_bvh_num_levels(bvh)), "host_build_s": sah_host_s, "device_build_s": sah_device_s,
                            "built_on_device": bool(on_device), "frame_ms": best}
    lib.lib.rt_bvh_destroy(bvh)
    return out_d


if __name__ == "__main__":
    main()
