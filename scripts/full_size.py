#!/usr/bin/env python3
"""BASELINE configs 2, 3 and 4 at FULL size through the one-call seam andi_hip_dist_matrix, on one GPU.

  c3: 109 genomes x 5.1 Mbp (C3-synth: d ~ U[1e-4, 5e-3]), Kimura -- BASELINE.json configs[2] on one GPU.
  c4: 3085 genomes x 2.1 Mbp (C4-synth, SURVEY.md 8d: d ~ U[1e-3, 1.5e-2]) -- the 3085 x 3085 matrix,
      386 batches of slot reuse, 3085 device suffix sorts, a 647 MB matrix.
  c5: 256 genomes x 50 Mbp (d ~ U[1e-3, 5e-2]) + 99 bootstrap matrices on the device.

Evidence written to gpurun_out/r06_<config>_full.json: wall-clock of the call, the ANDI_E2E_TRACE split (stderr of
the library, captured), sampled rows against the oracle, and the oracle's OpenMP port timed on a row sample on all
host cores (checker code: timed as a baseline only).  /root/reference is not needed.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def fast_set(n, length, d_lo, d_hi, seed, threads):
    """star set as synth.genome_set, but substitutions drawn per position (binomial count instead of an exact one):
    two orders of magnitude faster to generate at these sizes; same base, same kind of data"""
    from concurrent.futures import ThreadPoolExecutor
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = np.random.Generator(np.random.PCG64(seed)).integers(0, 4, size=length, dtype=np.uint8)
    ds = np.random.Generator(np.random.PCG64(seed ^ 0x5EED)).uniform(d_lo, d_hi, size=n)

    def one(k):
        rng = np.random.Generator(np.random.PCG64(seed + 1 + k))
        p = 0.75 - 0.75 * np.exp(-4.0 * ds[k] / 3.0)
        out = base.copy()
        # positions by geometric gaps: O(mutations), not O(length)
        m = int(p * length * 1.1) + 1000
        pos = np.cumsum(rng.geometric(p, size=m)) - 1
        pos = pos[pos < length]
        out[pos] = (out[pos] + rng.integers(1, 4, size=len(pos), dtype=np.uint8)) & 3
        return acgt[out].tobytes()

    with ThreadPoolExecutor(threads) as pool:
        return list(pool.map(one, range(n))), [float(x) for x in ds]


def cli_run(args, seqs, G, L, model, model_name, cores):
    """BASELINE's job as a user runs it: FASTA files in, PHYLIP matrix out (src/andi.c:63-394, src/io.c:159-322)."""
    import shutil
    import subprocess
    import andi_amd
    from concurrent.futures import ThreadPoolExecutor
    total = sum(len(s) for s in seqs)
    d = args.dir
    if not d:
        for cand in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
            try:
                if shutil.disk_usage(cand).free > 1.3 * total + (2 << 30):
                    d = cand
                    break
            except OSError:
                pass
    if not d:
        sys.exit("full_size.py --cli: no directory with room for %.1f GB of FASTA" % (total / 1e9))
    d = os.path.join(d, "andi_cli_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    names = ["g%04d" % k for k in range(G)]

    def write(k):
        s, path = seqs[k], os.path.join(d, names[k] + ".fasta")
        parts = [s]
        if args.contigs > 1:
            cut = sorted(set(int(x) for x in np.linspace(0, len(s), args.contigs + 1)))
            parts = [s[a:b] for a, b in zip(cut[:-1], cut[1:])]
        with open(path, "wb") as f:
            for c, part in enumerate(parts):
                f.write((">%s_%d some comment\n" % (names[k], c) if args.contigs > 1 else ">%s\n" % names[k]).encode())
                a = np.frombuffer(part, np.uint8)
                full = len(a) // 70 * 70
                lines = np.empty((full // 70, 71), np.uint8)
                lines[:, :70] = a[:full].reshape(-1, 70)
                lines[:, 70] = 10
                f.write(lines.tobytes())
                if full < len(a):
                    f.write(a[full:].tobytes() + b"\n")
        return path
    t0 = time.time()
    with ThreadPoolExecutor(min(cores, 32)) as pool:
        paths = list(pool.map(write, range(G)))
    t_write = time.time() - t0
    fof = os.path.join(d, "files.txt")
    with open(fof, "w") as f:
        f.write("\n".join(paths) + "\n")
    exe = os.path.join(ROOT, "andi_amd", "andi-hip")
    cmd = [exe, "--file-of-filenames", fof, "-m", model_name] + (["-j"] if args.contigs > 1 else [])
    out_path = os.path.join(d, "matrix.phy")
    try:
        t0 = time.time()
        with open(out_path, "wb") as fo:
            r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, ANDI_HIP_CLI_TRACE="1"), timeout=3000)
        wall = time.time() - t0
        err = r.stderr.decode(errors="replace")
        tr = [ln for ln in err.splitlines() if "andi-hip trace" in ln]
        import re
        m = re.search(r"ingest ([0-9.]+) s, matrix ([0-9.]+) s, print ([0-9.]+) s", tr[-1]) if tr else None
        ingest, matrix, prnt = (float(x) for x in m.groups()) if m else (None, None, None)
        printed = open(out_path, "rb").read().decode()
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # the same matrix through the seam, printed by the library's own formatter
    joined = [b"!".join(s[a:b] for a, b in zip(cut[:-1], cut[1:])) for s in seqs for cut in [sorted(set(int(x) for x in np.linspace(0, len(s), args.contigs + 1)))]] if args.contigs > 1 else seqs
    t0 = time.time()
    M = andi_amd.dist_matrix(joined, model=model)
    t_seam = time.time() - t0
    text, _, _ = andi_amd.format_distances(M, names, model)
    pairs = G * (G - 1)
    out = {
        "config": "%s-synth through the command line: %d FASTA files (%d nt each, 70-column lines%s), andi-hip --file-of-filenames -m %s%s, one GPU"
                  % (args.config.upper(), G, L, ", %d records per file" % args.contigs if args.contigs > 1 else "", model_name, " -j" if args.contigs > 1 else ""),
        "fasta_bytes": int(total + total // 70 + 16 * G), "files_written_s": t_write, "exit_code": r.returncode,
        "cli_wall_s": wall, "ingest_s": ingest, "matrix_s": matrix, "print_s": prnt,
        "ingest_plus_print_fraction_of_wall": (ingest + prnt) / wall if m else None,
        "pairs_per_s_cli_wall": pairs / wall, "seam_alone_s": t_seam,
        "output_bytes": len(printed), "output_equals_format_distances_of_the_seam_matrix": printed == text,
        "stderr_tail": err.splitlines()[-3:],
    }
    path = args.out or os.path.join(ROOT, "gpurun_out", "r07_cli_%s.json" % args.config)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=("c3", "c4", "c5"))
    ap.add_argument("--genomes", type=int, default=0)
    ap.add_argument("--length", type=int, default=0)
    ap.add_argument("--check-rows", type=int, default=3)
    ap.add_argument("--cpu-rows", type=int, default=2)
    ap.add_argument("--bootstrap", type=int, default=99)
    ap.add_argument("--low-memory", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--cli", action="store_true",
                    help="the same job through the command line: the genomes as FASTA files (one per genome, 70-column lines) in --dir, "
                         "`andi-hip --file-of-filenames` on them, ingest / matrix / print seconds (ANDI_HIP_CLI_TRACE), the printed matrix "
                         "against andi_hip_format_distances of the seam's own matrix")
    ap.add_argument("--contigs", type=int, default=0, help="--cli: every genome as that many records, joined again by -j (C4's assemblies)")
    ap.add_argument("--dir", default="", help="--cli: where the FASTA files go (default: /dev/shm if it has room, else $TMPDIR)")
    args = ap.parse_args()
    if args.config == "c3":
        G, L, dlo, dhi, model_name = args.genomes or 109, args.length or 5_100_000, 1e-4, 5e-3, "Kimura"
    elif args.config == "c4":
        G, L, dlo, dhi, model_name = args.genomes or 3085, args.length or 2_100_000, 1e-3, 1.5e-2, "JC"
    else:
        G, L, dlo, dhi, model_name = args.genomes or 256, args.length or 50_000_000, 1e-3, 5e-2, "JC"
    cores = os.cpu_count() or 1

    import andi_amd
    from andi_amd import lib
    from oracle import orc

    t0 = time.time()
    seqs, ds = fast_set(G, L, dlo, dhi, 1729, min(cores, 64))
    t_gen = time.time() - t0
    model = andi_amd.M_KIMURA if model_name == "Kimura" else andi_amd.M_JC
    if args.cli:
        return cli_run(args, seqs, G, L, model, model_name, cores)

    # the library's trace goes to the C stderr: capture it through a file
    os.environ["ANDI_E2E_TRACE"] = "1"
    andi_amd.lib.reload_knobs()
    tf = tempfile.TemporaryFile(mode="w+b")
    saved = os.dup(2)
    sys.stderr.flush()
    os.dup2(tf.fileno(), 2)
    t0 = time.time()
    try:
        M = andi_amd.dist_matrix(seqs, model=model, low_memory=args.low_memory)
    finally:
        os.dup2(saved, 2)
        os.close(saved)
    wall = time.time() - t0
    tf.seek(0)
    trace = tf.read().decode(errors="replace").strip().splitlines()

    pairs = G * (G - 1)
    out = {
        "config": "%s-synth full: %d genomes x %d nt, d~U[%g,%g] from a common base, %s, one GPU, andi_hip_dist_matrix%s"
                  % (args.config.upper(), G, L, dlo, dhi, model_name, " (low_memory)" if args.low_memory else ""),
        "pairs": pairs, "query_nt_scanned": pairs * L, "generate_s": t_gen,
        "dist_matrix_wall_s": wall, "pairs_per_s": pairs / wall,
        "algorithmic_GBps_whole_call": 2.0 * pairs * L / wall / 1e9,
        "trace": trace, "matrix_bytes": int(M.nbytes), "gather": lib.last_gather(),
    }
    diag_ok = all(int(M[i, i, 0]) == 9 and int(M[i, i, 16]) == 9 for i in range(0, G, max(1, G // 50)))
    out["diagonal_placeholders_ok"] = bool(diag_ok)

    # ---- sampled rows against the oracle; the same rows time the CPU port (one subject at a time, all cores on its queries)
    rows = sorted(set(int(x) for x in np.linspace(0, G - 1, max(args.check_rows, args.cpu_rows)).round()))[:max(args.check_rows, args.cpu_rows)]
    checked, cpu_build, cpu_scan, cpu_pairs = [], 0.0, 0.0, 0
    for k, i in enumerate(rows):
        t1 = time.time()
        O = orc.OracleEsa(seqs[i])
        t2 = time.time()
        row = orc.scan_row(O, seqs, i, model, threads=cores)
        t3 = time.time()
        O.close()
        same = bool((row == M[i]).all())
        checked.append({"row": i, "equal_to_oracle": same, "oracle_index_build_s": t2 - t1, "oracle_scan_s": t3 - t2})
        if k < args.cpu_rows:
            cpu_build += t2 - t1
            cpu_scan += t3 - t2
            cpu_pairs += G - 1
    out["rows_checked_against_oracle"] = checked
    out["parity"] = all(c["equal_to_oracle"] for c in checked)
    if cpu_pairs:
        per_row = (cpu_build + cpu_scan) / len(rows[:args.cpu_rows])
        out["cpu_baseline"] = {
            "kind": "port", "cores": cores, "unit": "pairs/s",
            "sample": "%d subject rows x %d queries: index built by one core (own suffix sorter, not libdivsufsort), "
                      "queries scanned by all %d cores (distMatrixLM's loop, src/dist_hack.h:59-60)" % (len(rows[:args.cpu_rows]), G - 1, cores),
            "value": cpu_pairs / (cpu_build + cpu_scan), "scan_only_pairs_per_s": cpu_pairs / cpu_scan if cpu_scan else None,
            "index_build_s_per_subject": cpu_build / len(rows[:args.cpu_rows]),
            "extrapolated_full_matrix_s": per_row * G,
            "extrapolated_full_matrix_s_scan_only": cpu_scan / len(rows[:args.cpu_rows]) * G,
        }
        out["gpu_over_cpu_wall"] = out["cpu_baseline"]["extrapolated_full_matrix_s"] / wall
        out["gpu_over_cpu_wall_cpu_scan_only"] = out["cpu_baseline"]["extrapolated_full_matrix_s_scan_only"] / wall

    if args.config == "c5" and args.bootstrap:
        ctx = andi_amd.Context(0)
        t0 = time.time()
        B = lib.bootstrap(ctx, M, args.bootstrap, seed=12345)
        tb = time.time() - t0
        tot = M[:, :, :16].astype(np.int64).sum(axis=2)
        iu = np.triu_indices(G, 1)
        # calculate_bootstrap resamples i < j and mirrors (src/process.c:289-321)
        okt = all(bool((B[r][:, :, :16].astype(np.int64).sum(axis=2)[iu] > 0).all()) for r in range(0, args.bootstrap, 10))
        out["bootstrap"] = {"replicates": args.bootstrap, "seconds": tb, "cells_nonempty": okt,
                            "mean_total_ratio": float(np.mean(B[0][:, :, :16].astype(np.int64).sum(axis=2)[iu] / np.maximum(1, (tot + tot.T)[iu])))}
        ctx.close()

    if model_name == "Kimura":  # the distances themselves for a few pairs (src/model.c:103-127): within 1e-9 of the oracle's -- they are equal
        i = rows[0]
        js = [j for j in (0, G // 2, G - 1) if j != i][:2]
        O = orc.OracleEsa(seqs[i])
        out["kimura_distances"] = []
        for j in js:
            Oj = orc.OracleEsa(seqs[j])
            want = orc.estimate(O.dist_anchor(seqs[j], model).astype(np.uint64) + Oj.dist_anchor(seqs[i], model), model)
            got = andi_amd.estimate(M[i, j].astype(np.uint64) + M[j, i], model)
            out["kimura_distances"].append({"pair": [i, j], "gpu": got, "oracle": want, "abs_diff": abs(got - want)})
            Oj.close()
        O.close()
    path = args.out or os.path.join(ROOT, "gpurun_out", "r06_%s_full.json" % args.config)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("config", "dist_matrix_wall_s", "pairs_per_s", "parity") if k in out}))
    for t in trace:
        print(t)


if __name__ == "__main__":
    main()
