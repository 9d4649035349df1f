#!/usr/bin/env python3
"""Benchmark of the anchor-distance hot path on MI355X.

Metric (BASELINE.json): ordered genome-pairs/sec over the N x N loop
(n^2 - n comparisons, the unit of the reference's progress counter,
src/dist_hack.h:41-42), plus the achieved HBM GB/s of the anchor-scan kernel
against the 8 TB/s roofline.

Workload at --gpus 1: BASELINE.json configs[1] as synthetic data ("C2-synth",
SURVEY.md §8d): 29 genomes of 4.9 Mbp, each diverged from a common base by
d_k ~ U[0.0004, 0.03], JC model.  One step = one pass of the device path over
the whole set: per subject the device index build (the scan index: packed text and
K-mer probe table, from the resident RS + suffix array), then the anchor scan of
every query against every subject (passes A/B of scan_lane.hip, pass C of
scan.hip), then (N > 1) the RCCL gather of the row blocks.  The texts and their
suffix arrays are staged before the timed region (the suffix arrays are built on
the device, sa_device.hip; the host sorter's time is reported beside it); the whole
job through the one-call seam andi_hip_dist_matrix -- everything included, cold -- is
reported under "end_to_end", never in "value".

N > 1 (one process per GPU, launched by torch.distributed.run): one G x G
matrix with G ~ 29*sqrt(N) genomes, rows block-partitioned over the ranks, so
per-GPU work stays that of one 29 x 29 set ("weak").  No data-path collective;
one gather at the end.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=0, help="genomes in the set (default: 29 at 1 GPU)")
    ap.add_argument("--subjects", type=int, default=0,
                    help="scan only the first S genomes as subjects (S rows x all genomes; default: all rows)")
    ap.add_argument("--length", type=int, default=4_900_000)
    ap.add_argument("--dlo", type=float, default=0.0004)
    ap.add_argument("--dhi", type=float, default=0.03)
    ap.add_argument("--segment", type=int, default=0)
    ap.add_argument("--set", choices=("star", "realistic", "tree"), default="star",
                    help="star: substitutions only (the reference's generator, test/test_fasta.cxx); realistic: repeats on "
                         "both strands, indels, inversions, unrelated islands (andi_amd/synth.py: realistic_set); tree: "
                         "substitutions along a random tree, pairwise distances 4.4e-4 ... 2.6e-2 (tree_set)")
    ap.add_argument("--model", choices=("raw", "jc", "kimura"), default="jc", help="the estimator's model (andi -m): what the counts are made for")
    ap.add_argument("--contigs", type=int, default=0, help="every genome cut into that many contigs joined by '!' (andi --join, src/sequence.c:78-125: "
                                                           "what multi-contig assemblies like BASELINE's Maela set look like to the scan)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads (structured genomes, tree-structured set)")
    ap.add_argument("--seam-child", default="", help=argparse.SUPPRESS)  # internal: the multi-GPU seam in a process of its own
    ap.add_argument("--seed", type=int, default=1729)
    return ap.parse_args()


def cpu_baseline(seqs, p_value, model):
    """The oracle (a port of the reference's OpenMP path) on this box's host cores.  Checker code: timed as a
    baseline only, never part of the product path.  Both of the reference's loops: subject-parallel (distMatrix,
    src/dist_hack.h:46-47; whole workload) and query-parallel (distMatrixLM, src/dist_hack.h:59-60; all host
    cores on the queries of one subject at a time; a sample of the subjects)."""
    from oracle import orc
    cores = os.cpu_count() or 1
    threads = min(cores, len(seqs))
    t0 = time.time()
    M, (t_build, t_scan) = orc.dist_matrix(seqs, p_value=p_value, model=model, threads=threads, times=True)
    wall = time.time() - t0
    n = len(seqs)
    # query-parallel flavour on a bounded sample: 3 subjects, every query, all cores
    sample = min(3, n)
    t_lm_build = t_lm_scan = 0.0
    lm_ok = True
    for i in range(sample):
        t1 = time.time()
        O = orc.OracleEsa(seqs[i], p_value)
        t2 = time.time()
        row = orc.scan_row(O, seqs, i, model, threads=cores)
        t3 = time.time()
        O.close()
        t_lm_build += t2 - t1
        t_lm_scan += t3 - t2
        lm_ok = lm_ok and bool((row == M[i]).all())
    # the same port fed suffix arrays from a linear-time sorter (the product's host SA-IS, andi_hip_suffix_array) instead
    # of the oracle's own multikey quicksort: what a libdivsufsort-based andi (src/esa.c:303) is closer to
    from concurrent.futures import ThreadPoolExecutor
    from andi_amd import lib as _lib

    def one_row(i):
        RS, _gc, _thr, sa = _lib.prepare_host(seqs[i], p_value)
        O = orc.OracleEsa(seqs[i], p_value, sa=sa)
        row = orc.scan_row(O, seqs, i, model, threads=1)
        O.close()
        return row
    t4 = time.time()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        rows = list(pool.map(one_row, range(n)))
    wall_fast = time.time() - t4
    fast_ok = all(bool((rows[i] == M[i]).all()) for i in range(n))
    return M, {
        "value": (n * n - n) / wall, "unit": "pairs/s", "cores": threads, "kind": "port",
        "with_fast_sorter": {
            "value": (n * n - n) / wall_fast, "unit": "pairs/s", "cores": threads, "wall_s": wall_fast, "equal": fast_ok,
            "note": "the same port with suffix arrays from a linear-time sorter (SA-IS, andi_hip_suffix_array) instead of the "
                    "oracle's multikey quicksort; a libdivsufsort-based andi (src/esa.c:303) is closer to THIS figure",
        },
        "sample": "whole workload: %d genomes, %d ordered pairs, index build (own suffix sorter, not "
                  "libdivsufsort) + scan, %.1f s wall" % (n, n * n - n, wall),
        "host_cores_available": cores,
        "index_build_core_s": t_build, "scan_core_s": t_scan,
        "scan_only_pairs_per_s_per_core": (n * n - n) / t_scan if t_scan > 0 else None,
        "query_parallel": {
            "note": "distMatrixLM-style (src/dist_hack.h:59-60): one subject at a time, its index built by one core, "
                    "its queries scanned by all %d host cores; sample = first %d subjects x %d queries" % (cores, sample, n - 1),
            "value": sample * (n - 1) / (t_lm_build + t_lm_scan), "unit": "pairs/s", "cores": cores,
            "scan_only_pairs_per_s": sample * (n - 1) / t_lm_scan if t_lm_scan > 0 else None,
            "index_build_s_per_subject": t_lm_build / sample, "equal_to_subject_parallel": lm_ok,
        },
    }


def pass_a_of(tm):
    """which kernel(s) ran pass A, and -- in calls routed per pair -- the fraction of the query nucleotides each took"""
    launches = max(int(tm["scan_launches"]), 1)
    if tm["routed_calls"]:
        c, l = float(tm["coop_query_nt"]), float(tm["lane_query_nt"])
        tot = max(c + l, 1.0)
        wk = "k_pool_cold" if int(tm.get("pool_calls", 0)) >= max(int(tm["routed_calls"]), 1) else "k_coop_cold"  # which wavefront kernel the calls chose
        name = wk if l == 0 else ("k_lane_cold" if c == 0 else wk + "+lanes")
        return name, {wk: c / tot, "lanes (k_lane_cold, k_lane_quad)": l / tot,
                      "pairs_handed_back_per_call": int(tm["coop_fallbacks"]) / max(int(tm["routed_calls"]), 1)}
    return (("k_pool_cold" if int(tm.get("pool_calls", 0)) >= launches else "k_coop_cold") if tm["coop_calls"] >= launches else "k_lane_cold"), None


MODELS = {"raw": 0, "jc": 1, "kimura": 2}  # andi_amd.M_RAW, M_JC, M_KIMURA (include/andi_hip.h)
MODEL_NAMES = {0: "RAW", 1: "JC", 2: "Kimura"}


def make_set(kind, G, length, dlo, dhi, seed):
    from andi_amd import synth
    if kind == "fast":  # hundreds of genomes (BASELINE's config 2): substitutions drawn position by position, by a pool of threads
        return synth.genome_set_fast(G, length, dlo, dhi, seed=seed, threads=min(os.cpu_count() or 1, 32))[0]
    if kind == "realistic":
        return synth.realistic_set(G, length, dlo, dhi, seed=seed)[0]
    if kind == "tree":
        return synth.tree_set(G, length, seed=seed)[0]
    return synth.genome_set(G, length, dlo, dhi, seed=seed)[0]


def workload_name(kind, G, S, length, dlo, dhi, seed, world, model=1):
    """what the set really is: the C2 name only for C2's shape (29 genomes of 4.9 Mbp per GPU tile)"""
    import andi_amd.shard as shard
    c2 = length == 4_900_000 and G == shard.weak_scaling_set_size(world) and (kind != "star" or (dlo, dhi) == (0.0004, 0.03))
    c4 = kind == "star" and G == 3085 and length == 2_100_000  # BASELINE's config 3 as synthetic data (C4-synth)
    c3 = kind == "fast" and G == 109 and length == 5_100_000   # BASELINE's config 2 (109 E. coli ST131, Kimura) as synthetic data: C3-synth
    tag = {"star": "synth", "fast": "synth", "realistic": "realistic", "tree": "tree"}[kind]
    what = {"star": "d~U[%g,%g] from a common base (star)" % (dlo, dhi),
            "fast": "d~U[%g,%g] from a common base (star; substitutions drawn per position)" % (dlo, dhi),
            "realistic": "d~U[%g,%g] from a common base, with repeats, indels, inversions, 10%% unrelated sequence" % (dlo, dhi),
            "tree": "substitutions along a random tree, pairwise d 4.4e-4 ... 2.6e-2"}[kind]
    return "%s: %d genomes x %d nt, %s, %s, seed %d; %s rows block-partitioned over %d GPU(s)" % (
        ("C2-" + tag) if c2 else ("C4-synth" if c4 else "C3-synth" if c3 else "synthetic " + tag + " set"), G, length, what,
        MODEL_NAMES[model], seed, "all" if S == G else "the first %d subject" % S, world)


def secondary(kind, args, model, p_value, G=29, L=None, S=None, dlo=None, dhi=None, seam=False, contigs=0):
    """The same step on another set (one GPU, after the headline's timed region): structured genomes, the tree-structured
    variant, and a call of BASELINE's config 3 (S subject rows of the 3085-genome set: the shape the north_star's target
    is stated on) -- reported beside, never instead of, the star headline."""
    import andi_amd
    from andi_amd import lib
    L = L or args.length
    S = S or G
    dlo = args.dlo if dlo is None else dlo
    dhi = args.dhi if dhi is None else dhi
    seqs = make_set(kind, G, L, dlo, dhi, args.seed)
    if contigs > 1:  # andi --join: every genome a set of contigs, cut at places of its own
        from andi_amd import synth
        seqs = [synth.join_contigs(sq, contigs, seed=args.seed + 7 * k) for k, sq in enumerate(seqs)]
    ctx = andi_amd.Context(0)
    ctx.expect_queries(G - 1)
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, s, p_value, build=False, sa="device") for s in seqs[:S]]
    M = ctx.alloc(S * G * 68)
    selfs = list(range(S))

    def step():
        lib.build_indexes(ctx, esas)
        lib.scan_rows_dev(ctx, esas, selfs, Q, model, args.segment, M)
        ctx.sync()

    step()
    ctx.timings_reset()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    el = time.perf_counter() - t0
    tm = ctx.timings()
    scan_ms = tm["scan_ms"] / max(int(tm["scan_launches"]), 1)
    alg = 2.0 * tm["scan_query_nt"] / max(int(tm["scan_launches"]), 1)
    out = {"workload": workload_name(kind, G, S, L, dlo, dhi, args.seed, 1, model) + (", every genome as %d contigs joined by '!' (--join)" % contigs if contigs > 1 else ""),
           "pairs_per_s": S * (G - 1) / (el / steps), "ms_per_step": 1e3 * el / steps,
           "roofline_frac": alg / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if scan_ms > 0 else None,
           "pass_a_kernel": pass_a_of(tm)[0], "pass_a_query_nt_fraction": pass_a_of(tm)[1],
           "index_build_ms": tm["build_ms"] / steps, "scan_cold_pass_ms": tm["scan_ms"] / steps,
           "scan_stitch_reduce_ms": tm["stitch_ms"] / steps, "fixups_per_step": int(tm["fixups"]) // steps}
    if seam:  # the same job once through the one-call seam (everything included), and its matrix against the step's
        import numpy as np
        step_M = np.empty((S, G, 17), np.uint32)
        ctx._check(lib.load().andi_hip_copy_to_host(ctx._h, step_M.ctypes.data, M, step_M.nbytes), "copy_to_host")
    ctx.free(M)
    for e in esas:
        e.close()
    Q.close()
    ctx.close()
    if seam:
        t0 = time.time()
        M1 = andi_amd.dist_matrix(seqs, p_value=p_value, model=model)
        out["dist_matrix_e2e_s"] = time.time() - t0
        out["dist_matrix_pairs_per_s"] = G * (G - 1) / out["dist_matrix_e2e_s"]
        out["dist_matrix_equals_step"] = bool((M1[:S] == step_M).all())
        i, j = 0, G - 1
        out["sample_distance"] = andi_amd.estimate(M1[i, j].astype(np.uint64) + M1[j, i], model)
    return out


def c3_strong(args, p_value, world, rank, local_rank, dist, torch):
    """BASELINE's configs[2] as it is written -- 109 genomes of 5.1 Mbp, Kimura, 'pair-tile sharding 1 -> 8': the SAME matrix
    whatever the number of GPUs (strong scaling), its rows block-partitioned over the ranks, one gather per step; all ranks
    call this after the headline's timed region.  Returns the record on rank 0."""
    import numpy as np
    import andi_amd
    from andi_amd import lib, shard
    G, L, dlo, dhi, model = 109, 5_100_000, 1e-4, 5e-3, MODELS["kimura"]
    seqs = make_set("fast", G, L, dlo, dhi, args.seed)
    r0, r1 = shard.row_block(G, world, rank)
    ctx = andi_amd.Context(local_rank)
    ctx.expect_queries(G - 1)
    Q = andi_amd.Queries(ctx, seqs)
    esas = [andi_amd.Esa(ctx, seqs[i], p_value, build=False, sa="device") for i in range(r0, r1)]
    block = torch.zeros((shard.max_rows(G, world), G, 17), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    dptr = andi_amd.lib._P(block.data_ptr())
    selfs = list(range(r0, r1))
    gathered = [None]

    def step():
        lib.build_indexes(ctx, esas)
        lib.scan_rows_dev(ctx, esas, selfs, Q, model, 0, dptr)
        ctx.sync()
        gathered[0] = shard.gather_matrix(block, G, dist, world, rank, force=dist is not None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    fence()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    for e in esas:
        e.close()
    Q.close()
    ctx.close()
    if rank != 0:
        return None, None, None
    full = gathered[0]
    rec = {"workload": workload_name("fast", G, G, L, dlo, dhi, args.seed, world, model), "scaling": "strong",
           "n_gpus": world, "ms_per_step": 1e3 * el / steps, "pairs_per_s": G * (G - 1) / (el / steps),
           "sample_distance": andi_amd.estimate(full[0, G - 1].astype(np.uint64) + full[G - 1, 0], model)}
    return rec, full, ("fast", G, L, dlo, dhi, model)


def seam_child(spec):
    """--seam-child: the whole job through the product's one-call seam on all GPUs of the node -- andi_hip_dist_matrix with
    num_gpus = N: a driver thread and a context per device, the RCCL gather of the row blocks behind the C-ABI -- in a
    process of its own (the library loads ROCm's librccl, which must not meet the copy PyTorch carries)."""
    import numpy as np
    import andi_amd
    from andi_amd import lib
    kind, G, length, dlo, dhi, seed, gpus, path, model = spec.split(",")
    seqs = make_set(kind, int(G), int(length), float(dlo), float(dhi), int(seed))
    out = {"gpus_visible": lib.device_count()}
    try:
        t0 = time.time()
        M = andi_amd.dist_matrix(seqs, model=int(model), num_gpus=int(gpus))
        out["seam_multi_gpu_s"] = time.time() - t0
        out["gather"] = lib.last_gather()
        np.save(path, M)
    except Exception as e:  # the exact error string is the evidence
        out["error"] = str(e)
    print("SEAM " + json.dumps(out))


def main():
    args = parse()
    if args.seam_child:
        return seam_child(args.seam_child)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if max(args.gpus, 1) != world:
        raise SystemExit("bench.py --gpus %d needs %d ranks: launch it as `python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...` (one process per GPU); "
                         "WORLD_SIZE is %d" % (args.gpus, args.gpus, args.gpus, args.gpus, world))

    import numpy as np
    import torch

    import andi_amd
    from andi_amd import lib, shard, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the anchor-distance engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run also with one rank
    # The RCCL process group is initialised AFTER the engine has allocated what it works on (below, behind one untimed
    # pass): device memory allocated once RCCL is up makes pass A -- scattered, latency-bound loads -- 9 % slower
    # (6.05 -> 6.57 ms, measured with one rank; ANDI_BENCH_EARLY_INIT=1 restores that order).
    # andi_hip_dist_matrix creates its communicators after the scans for the same reason.
    late_init = not os.environ.get("ANDI_BENCH_EARLY_INIT")
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not late_init:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    G = args.genomes or shard.weak_scaling_set_size(world)
    model = MODELS[args.model]
    p_value = 0.025
    t_gen = time.time()
    seqs = make_set(args.set, G, args.length, args.dlo, args.dhi, args.seed)
    if args.contigs > 1:  # the same genomes as joined contigs: separators in subjects and queries
        seqs = [synth.join_contigs(sq, args.contigs, seed=args.seed + 7 * k) for k, sq in enumerate(seqs)]
    t_gen = time.time() - t_gen

    S = args.subjects if 0 < args.subjects <= G else G  # subject rows of the job
    r0, r1 = shard.row_block(S, world, rank)
    ctx = andi_amd.Context(local_rank)
    ctx.expect_queries(G - 1)  # as andi_hip_dist_matrix does: every subject meets every other sequence
    # ---- untimed staging: queries, and for the owned rows RS (host: seq_subject_init) and its suffix array (device)
    t_stage = time.time()
    Q = andi_amd.Queries(ctx, seqs)
    from concurrent.futures import ThreadPoolExecutor
    host_threads = max(1, min((os.cpu_count() or 1) // max(world, 1), r1 - r0))
    with ThreadPoolExecutor(host_threads) as pool:  # RS = revcomp(S) # S, gc, threshold on the host cores
        prepared = list(pool.map(lambda i: lib.subject_prepare(seqs[i], p_value), range(r0, r1)))
    ctx.timings_reset()
    esas = [andi_amd.Esa(ctx, seqs[i], p_value, build=False, prepared=prepared[i - r0] + ("device",)) for i in range(r0, r1)]
    del prepared
    ctx.sync()
    t_stage = time.time() - t_stage
    sa_ms = ctx.timings()["sa_ms"]
    nsub = r1 - r0
    block = torch.zeros((shard.max_rows(S, world), G, 17), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # the engine writes the block from its own stream: the fill must have landed
    gathered = [None]
    selfs = list(range(r0, r1))
    dptr = andi_amd.lib._P(block.data_ptr())

    def step():
        lib.build_indexes(ctx, esas)  # device index builds (probe tables), all subjects in one pair of launches
        lib.scan_rows_dev(ctx, esas, selfs, Q, model, args.segment, dptr)  # anchor scan
        ctx.sync()  # the engine's stream is not torch's: finish before the collective
        if use_dist:  # RCCL over xGMI: the one exchange of the job, 68 B per ordered pair
            gathered[0] = shard.gather_matrix(block, G, dist, world, rank, force=True, rows=S)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if use_dist and late_init:  # everything the engine allocates exists before RCCL is initialised (see above)
        lib.build_indexes(ctx, esas)
        lib.scan_rows_dev(ctx, esas, selfs, Q, model, args.segment, dptr)
        ctx.sync()
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    for _ in range(args.warmup):
        step()
    fence()
    ctx.timings_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tm = ctx.timings()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_total = S * (G - 1)
    ms_per_step = 1e3 * elapsed / args.steps
    value = pairs_total / (elapsed / args.steps)

    # roofline of the dominant kernel (pass A of the scan) from HIP events on
    # the engine's own stream: algorithmic bytes = 2 * query length per pair
    launches = max(int(tm["scan_launches"]), 1)
    scan_ms = tm["scan_ms"] / launches
    alg_bytes = 2.0 * tm["scan_query_nt"] / launches  # (query lengths differ in the realistic sets: the sum over the scanned pairs)
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic, traffic_source = None, None
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        try:
            rec = json.load(open(prof))
            key = "G%d_L%d_seg%d" % (G, args.length, args.segment)
            traffic = rec.get(key, {}).get("hbm_bytes_per_launch")  # rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE
            if traffic is not None:  # a constant from a counter run of this workload, NOT of this run: say where it is from
                traffic_source = "profiles/traffic.json: %s (%s) -- rocprofv3 --pmc passes of the same command, corrected as " \
                                 "MI355X_MICROARCH.md prescribes; a recorded figure, not measured by this run" % (
                                     rec.get(key, {}).get("source", "counter summary under profiles/"), rec.get(key, {}).get("commit", "commit not recorded"))
        except Exception:
            traffic = None

    # measured device-copy ceiling beside the nominal peak: 1 GiB D2D, read + write bytes
    # (the engine's own 16-byte streaming kernel, HIP events on its stream -- torch's int32 copy_ measured 4.75 TB/s where
    # the guide's float4 copy does 6.29)
    copy_gbps = None
    if rank == 0:
        try:
            copy_gbps = lib.copy_ceiling(ctx, 1 << 30, 5)
        except Exception:
            copy_gbps = None

    # N > 1: the roofline of the slowest rank's launches (every rank runs the same kernel on its own rows)
    frac = achieved / HBM_PEAK_GBPS
    roofline_ranks = None
    if use_dist:
        t = torch.tensor([frac, -frac, scan_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        roofline_ranks = {"frac_best_rank": float(t[0].item()), "frac_worst_rank": -float(t[1].item()), "avg_launch_ms_slowest_rank": float(t[2].item())}
        frac = roofline_ranks["frac_worst_rank"]
        achieved = frac * HBM_PEAK_GBPS
    # pass A's kernel: one wavefront per chain (scan_coop.hip) where the call suits it, else one lane per chain
    scan_kernel, routed_fraction = pass_a_of(tm)
    out = None
    if rank == 0:
        full = gathered[0] if use_dist else shard.gather_matrix(block, G, rows=S)
        dmat = [andi_amd.estimate(full[0, j].astype(np.uint64) + full[j, 0], model) for j in range(1, min(S, 4))]
        out = {
            "metric": "genome-pairs/sec (ordered pairs, n^2-n) over the N x N anchor-distance loop",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_name(args.set, G, S, args.length, args.dlo, args.dhi, args.seed, world, model) + (", every genome as %d contigs joined by '!' (--join)" % args.contigs if args.contigs > 1 else ""),
                       "genomes": G, "subjects": S, "length": args.length, "model": MODEL_NAMES[model], "pairs": pairs_total,
                       "segment": args.segment or ("auto (pass A routed per pair: by wavefronts on segments of 32768 ... 524288 symbols (shorter in small calls), by lanes on segments chosen per pair, 2048 ... 16384)" if tm["routed_calls"] else "auto (chosen per pair from its sampled match lengths: 2048 ... 16384)")},
            "roofline": {"bound": "hbm", "kernel": scan_kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": frac, "traffic": traffic, "traffic_source": traffic_source,
                         "ranks": roofline_ranks,
                         "measured_copy_GBps": copy_gbps,
                         "measured_copy_kernel": "andi_hip_copy_ceiling: 16 bytes per lane, non-temporal, 1 GiB, 5 passes, read + written bytes",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_ms,
                         "launches": int(tm["scan_launches"]),
                         "note": "the dominant kernel: pass A alone, HIP events around its launch(es) on the engine's streams; the sampling "
                                 "and routing of the pairs, the layout and passes B/C are breakdown_ms_per_step.scan_stitch_reduce"},
            "breakdown_ms_per_step": {"index_build": tm["build_ms"] / args.steps,
                                      "scan_cold_pass": tm["scan_ms"] / args.steps,
                                      "scan_stitch_reduce": tm["stitch_ms"] / args.steps,
                                      "fixups": int(tm["fixups"]),
                                      "scan_calls_pass_a_by_wavefronts": int(tm["coop_calls"]),
                                      "scan_calls_routed_per_pair": int(tm["routed_calls"]),
                                      "pass_a_query_nt_fraction": routed_fraction,
                                      "scan_calls_with_per_pair_segments": int(tm["adaptive_calls"]),
                                      "scan_calls_with_one_segment_length": int(tm["uniform_calls"])},
            "end_to_end": {"note": "rank 0, untimed staging of its rows: RS on %d host threads, H2D of RS, suffix arrays "
                                   "on the device (sa_device.hip)" % host_threads,
                           "staging_s": t_stage, "device_suffix_sort_s": sa_ms * 1e-3, "generate_s": t_gen,
                           "pairs_per_s_incl_staging": (nsub * (G - 1)) / (t_stage + elapsed / args.steps)},
            "sample_distances": dmat,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        Mcpu, base = cpu_baseline(seqs, p_value, model)
        out["cpu_baseline"] = base
        out["parity_vs_cpu_baseline"] = bool((Mcpu[:S] == full).all())
    elif rank == 0 and not args.no_cpu_baseline:
        # N > 1: the baseline is carried forward on a bounded sample -- the first 29 genomes of the set, which ARE the
        # N = 1 workload (same seed, same base, same divergences: one 29 x 29 tile, what every GPU of the weak-scaling job does);
        # the other ranks wait at the barrier below.  Its counts check the gathered matrix's corner.
        tile = min(G, shard.weak_scaling_set_size(1))
        Mcpu, base = cpu_baseline(seqs[:tile], p_value, model)
        base["sample"] = "bounded sample at N > 1: the first %d genomes of the set = the N = 1 workload (one tile of the weak-scaling job); %s" % (tile, base["sample"])
        out["cpu_baseline"] = base
        out["parity_vs_cpu_baseline"] = bool((Mcpu[:min(S, tile), :tile] == full[:tile, :tile]).all()) if S >= tile else None
    elif rank == 0:
        out["cpu_baseline"] = None

    for e in esas:
        e.close()
    Q.close()
    ctx.close()
    if rank == 0 and world == 1 and S == G and not args.no_cpu_baseline:
        # the whole job through the one-call seam (context, staging, suffix arrays, index builds, scans, D2H), cold
        t0 = time.time()
        M1 = andi_amd.dist_matrix(seqs, p_value=p_value, model=model)
        e2e = time.time() - t0
        t0 = time.time()
        M1w = andi_amd.dist_matrix(seqs, p_value=p_value, model=model)  # the same call again: arena chunks, pinned buffers, code objects are there
        e2e_warm = time.time() - t0
        # once more with the library's own split of the call (ANDI_E2E_TRACE: the driver threads time their stages, with a
        # device wait behind each -- a little slower than the calls above): where the time goes on THIS box.  The host is
        # shared with other jobs (load average below); the stages that run on host cores feel it.
        traced = None
        try:
            import tempfile
            os.environ["ANDI_E2E_TRACE"] = "1"
            andi_amd.lib.reload_knobs()
            sys.stderr.flush()
            saved, tmp = os.dup(2), tempfile.TemporaryFile()
            os.dup2(tmp.fileno(), 2)
            try:
                t0 = time.time()
                andi_amd.dist_matrix(seqs, p_value=p_value, model=model)
                dt = time.time() - t0
            finally:
                os.dup2(saved, 2)
                os.close(saved)
            tmp.seek(0)
            lines = [ln.strip() for ln in tmp.read().decode(errors="replace").splitlines() if "andi_hip_dist_matrix trace" in ln]
            traced = {"wall_s": dt, "split": lines}
        except Exception as ex:  # diagnostics only
            traced = {"error": str(ex)}
        finally:
            os.environ.pop("ANDI_E2E_TRACE", None)
            andi_amd.lib.reload_knobs()
        t0 = time.time()
        M2 = andi_amd.dist_matrix(seqs, p_value=p_value, model=model, sa_on_host=True)
        e2e_host = time.time() - t0

        def seam_ms(tr, what):  # a figure of the traced call's split, in seconds
            import re
            for ln in (tr or {}).get("split", []):
                m = re.search(re.escape(what) + r" ([0-9.]+)", ln)
                if m:
                    return float(m.group(1)) * 1e-3
            return None
        out["end_to_end"].update({
            "dist_matrix_e2e_s": e2e, "dist_matrix_pairs_per_s": pairs_total / e2e,
            "dist_matrix_e2e_warm_s": e2e_warm, "dist_matrix_warm_pairs_per_s": pairs_total / e2e_warm,
            "dist_matrix_traced_call": traced, "host_load_average": list(os.getloadavg()),
            # the sorter inside the seam (its staging thread sorts back to back; device_suffix_sort_s above is the sum over the bench's own
            # staging loop, one subject at a time with host work between them) and what the scan thread waited for it
            "dist_matrix_suffix_sort_s": seam_ms(traced, "suffix arrays"), "dist_matrix_scan_wait_s": seam_ms(traced, "waiting for staged subjects"),
            "dist_matrix_e2e_s_suffix_arrays_on_host": e2e_host, "host_cores": os.cpu_count(),
            "dist_matrix_equals_step": bool((M1 == full).all() and (M1w == full).all() and (M2 == full).all())})
    def seam_on_all_gpus(kind, G_, L_, dlo_, dhi_, model_, full_):
        """the product's own multi-GPU path (api.hip: andi_hip_dist_matrix with num_gpus = N, RCCL gather behind the C-ABI) on a
        set, in a child process (a subprocess -- never a re-exec), after the timed region; the other ranks wait at the barrier"""
        import subprocess
        import tempfile
        path = os.path.join(tempfile.gettempdir(), "andi_seam_%d.npy" % os.getpid())
        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
               and not k.startswith("TORCHELASTIC") and not k.startswith("TORCH_NCCL")}
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        spec = ",".join(str(x) for x in (kind, G_, L_, dlo_, dhi_, args.seed, world, path, model_))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--seam-child", spec], env=env, cwd=ROOT,
                               capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("SEAM ")]
            res = json.loads(line[-1][5:]) if line else {"error": "no result; rc %d; %s" % (r.returncode, r.stderr[-600:])}
            if os.path.exists(path):
                res["equals_gathered_matrix"] = bool((np.load(path)[:len(full_)] == full_).all())
                os.remove(path)
        except Exception as e:
            res = {"error": repr(e)}
        return res

    if rank == 0 and world > 1:
        out["end_to_end"]["seam_multi_gpu"] = seam_on_all_gpus(args.set, G, args.length, args.dlo, args.dhi, model, full)
    strong = world > 1 or os.environ.get("ANDI_BENCH_C3_STRONG")  # (the variable: the code path on one GPU, for testing)
    if strong and args.set == "star" and not args.subjects and not args.genomes:
        # BASELINE's configs[2] itself: the same 109-genome Kimura matrix on the N ranks, and through the seam's own tiler
        rec, c3_full, c3_spec = c3_strong(args, p_value, world, rank, local_rank, dist if use_dist else None, torch)
        if rank == 0:
            rec["seam_multi_gpu"] = seam_on_all_gpus(*c3_spec, c3_full)
            out.setdefault("extra", {})["c3_strong"] = rec
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world == 1 and not args.no_extra and args.set == "star" and not args.subjects and not args.contigs:
        out.setdefault("extra", {}).update({
            "realistic": secondary("realistic", args, model, p_value),
            "tree_structured": secondary("tree", args, model, p_value),
            "c4_shape": secondary("star", args, model, p_value, G=3085, L=2_100_000, S=8, dlo=0.001, dhi=0.015),
            # ... and as what the Maela assemblies are: multi-contig drafts under --join (40 contigs per genome, cut at places of their own)
            "c4_shape_joined": secondary("star", args, model, p_value, G=3085, L=2_100_000, S=8, dlo=0.001, dhi=0.015, contigs=40),
            # BASELINE's configs[2] at full size on one GPU: the step, and the whole job once through the one-call seam
            "c3": secondary("fast", args, MODELS["kimura"], p_value, G=109, L=5_100_000, dlo=1e-4, dhi=5e-3, seam=True)})
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
