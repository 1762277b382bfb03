#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the FM-index search hot path on MI355X.

Headline workload (BASELINE.json configs[1]): a GRCh38-sized index (3.1 Gbp; synthetic genome, because there is
no network for the real one -- see DESIGN.md) and 10 M synthetic 100 bp single-end reads per GPU at
0.2 % substitutions.  One "step" = one pass of bwa_cal_sa_reg_gap over the whole batch, inputs and the
index resident in HBM.  With --gpus N every rank holds a replica of the index and its own shard of reads
(no collective on the data path; weak scaling).  `python bench.py --gpus N` without a launcher around it starts
its N ranks itself (a torch.distributed.run child, before anything touches a GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0 with `roofline` (algorithmic Occ-bucket bytes / HIP-event kernel time
vs. the 8 TB/s HBM peak), `cpu_baseline` (the reference's own compiled code, oracle/_ref, on the
host cores of this box, on a bounded sample of the same reads), `e2e` (BAM records in -> BAM records out) and, at full
size, `workloads`: BASELINE configs 5 (--adna) and 3 (--pe) and the repeat-family genome (--repeats) after the headline
steps, each with its own value / roofline / cpu_baseline / bit-exact sample.
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
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRCH38_LEN = 3_099_734_149   # GRCh38 primary assembly, total bases
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-len", type=int, default=int(os.environ.get("NABWA_BENCH_GENOME", GRCH38_LEN)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("NABWA_BENCH_READS", 10_000_000)))
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--sub-ppm", type=int, default=2000)
    ap.add_argument("--indel-ppm", type=int, default=0, help="per-base indel rate of the synthetic reads (not part of the headline workload)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--adna", action="store_true", help="SURVEY 8d config C5 instead of the headline workload: reads of 50-76 bases with "
                    "terminal deamination (5' C>T, 3' G>A, 30 %% decaying by 0.7 per base) + 1 %% substitutions, searched with -n 0.01 -o 2 -l 16500")
    ap.add_argument("--repeats", action="store_true", help="the genome with GRCh38-like repeat families (LINE-like, Alu-like with a young subfamily, tandem repeats; "
                    "synth_index.hip) instead of the uniform one with 2000 planted 5 kb repeats")
    ap.add_argument("--pe", action="store_true", help="BASELINE config 3 instead of the headline workload: 2 x 150 bp pairs at 2 %% substitutions, 10 %% of the reads with "
                    "a 1-base indel, inserts ~ N(400, 40): FM search of both ends + posn_pair + insert-size estimate + finish_pair; value = pairs/s")
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("NABWA_BENCH_PAIRS", 1_000_000)))
    ap.add_argument("--no-e2e", action="store_true", help="skip the BAM-records-in -> BAM-records-out leg of the headline run")
    ap.add_argument("--e2e-reads", type=int, default=1_000_000)
    ap.add_argument("--pipeline", action="store_true", help="steps alternate between two device-resident batches on two streams (not the headline mode)")
    ap.add_argument("--extras", choices=["auto", "on", "off"], default="auto", help="after the headline steps also run configs 5 (--adna), 3 (--pe) and the repeat-family "
                    "genome and attach them under `workloads` (auto: at the full genome size only)")
    ap.add_argument("--extra-steps", type=int, default=2, help="timed steps of each extra workload (one warm-up step before them)")
    ap.add_argument("--extra-adna-reads", type=int, default=6_250_000, help="config 5's share of one GPU: 50 M reads over 8 GPUs")
    ap.add_argument("--extra-repeat-reads", type=int, default=10_000_000)
    ap.add_argument("--extra-cpu-seconds", type=float, default=8.0)
    return ap.parse_args()


def launch_ranks(args):
    """`bench.py --gpus N` run directly: start the N ranks as a child `python -m torch.distributed.run` (one process per GPU over RCCL)
    and pass rank 0's line through.  Nothing in THIS process has touched a GPU (torch is not even imported), so no process that
    initialised HIP is ever replaced; the child's exit code is ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:])))
    return subprocess.run(cmd, env=env).returncode


class Ctx:
    pass


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    ctx = Ctx()
    ctx.args = args
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    ctx.torch = torch
    ctx.dist = None
    ctx.backend = os.environ.get("NABWA_BENCH_BACKEND", "nccl")          # "gloo": rehearsal of the N>1 path on one GPU
    one_dev = os.environ.get("NABWA_BENCH_SINGLE_DEVICE") == "1"
    ctx.dev = dev = local_rank if (world > 1 and not one_dev) else 0
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctx.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(ctx.backend)
        ctx.dist = dist

    ctx.nabwa = nabwa = importlib.import_module("network-aware-bwa_amd")
    ctx.synth = synth = importlib.import_module("network-aware-bwa_amd.synth")
    if not os.path.exists(nabwa.LIB_PATH) or not os.path.exists(synth.LIB_PATH):
        nabwa.build()
    import nabwa_testlib as T
    ctx.T = T
    ctx.want_cpu = rank == 0 and not args.no_cpu
    # a CPU leg is taken on rank 0 at N = 1 only (at N > 1 the host cores are the ranks'); the bit-exact sample check stays on at any N
    t_start = time.time()

    single = args.pe or args.adna or args.repeats
    extras = [] if (single or args.pipeline or quick_env() or args.extras == "off" or (args.extras == "auto" and args.genome_len != GRCH38_LEN)) \
        else ["adna", "pe", "repeats"]
    want_e2e = not args.no_e2e and not args.adna and not args.pe and not quick_env()
    need_ref = args.pe or want_e2e or "pe" in extras
    G = build_genome(ctx, args.genome_len, args.repeats, need_ref)
    if args.pe:
        out = pe_workload(ctx, G, args.pairs, args.steps, args.warmup, args.cpu_seconds)
    else:
        kind = "adna" if args.adna else ("repeats" if args.repeats else "headline")
        out = se_workload(ctx, G, kind, args.reads, args.steps, args.warmup, True, args.cpu_seconds, want_e2e)
    if extras:
        wl = {}
        es, ecpu = args.extra_steps, args.extra_cpu_seconds
        t0 = time.time()
        wl["adna"] = se_workload(ctx, G, "adna", args.extra_adna_reads, es, 1, False, ecpu, False)
        if rank == 0:
            log("extra workload adna: %.1f s" % (time.time() - t0))
        t0 = time.time()
        wl["pe"] = pe_workload(ctx, G, args.pairs, es, 1, ecpu)
        if rank == 0:
            log("extra workload pe: %.1f s" % (time.time() - t0))
        t0 = time.time()
        G.close()
        G = build_genome(ctx, args.genome_len, True, False)
        wl["repeats"] = se_workload(ctx, G, "repeats", args.extra_repeat_reads, es, 1, False, ecpu, False)
        if rank == 0:
            log("extra workload repeats: %.1f s" % (time.time() - t0))
            out["workloads"] = wl
    G.close()
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    if rank == 0 and out is not None:
        out["bench_wall_s"] = round(time.time() - t_start, 1)
        print(json.dumps(out), flush=True)


class Genome:
    """the synthetic text on the device, both FM-indexes in HBM (nabwa.Index), and -- for the CPU legs -- the index arrays on the host"""

    def close(self):
        if self.d_text is not None:
            self.d_text.free()
            self.d_text = None
        if self.ix is not None:
            self.ix.close()
            self.ix = None
        self.host_bwt = self.host_sa = self.pac = None


def build_genome(ctx, n, repeats, need_ref):
    nabwa, synth, dev, rank = ctx.nabwa, ctx.synth, ctx.dev, ctx.rank
    t0 = time.time()
    G = Genome()
    G.n, G.repeats = n, repeats
    G.d_text = synth.synth_text_repeats(n, 20261004, device=dev) if repeats else synth.synth_text(n, 20261004, n_dup=2000, dup_len=5000, device=dev)
    # with the SA samples (.sa/.rsa content): the index derives its full SA / inverse / text from them (text mode)
    parts = [synth.build_index(G.d_text, n, rev, 32, True, device=dev, verbose=(rank == 0)) for rev in (0, 1)]
    G.ix = nabwa.Index.from_arrays((parts[0][0].ptr, parts[0][1]), (parts[1][0].ptr, parts[1][1]),
                                   (parts[0][2].ptr, parts[0][3]), (parts[1][2].ptr, parts[1][3]), device=dev, device_ptrs=True)
    if rank == 0:
        log("index (%s genome): %d bp x2 FM-indexes built on GPU + re-packed in %.1f s (%.2f GB in HBM)"
            % ("repeat-family" if repeats else "uniform", n, time.time() - t0, G.ix.device_bytes() / 1e9))
    G.host_bwt = G.host_sa = G.pac = None
    if ctx.want_cpu:
        G.host_bwt = [p[0].to_host(np.uint32, p[1]) for p in parts]
        if need_ref:
            G.host_sa = [p[2].to_host(np.uint32, p[3]) for p in parts]
    for p in parts:
        p[0].free()
        p[2].free()
    if need_ref:
        # the finishing chains read the packed reference (.pac layout) and the contigs' annotation from the host
        G.pac = pack_text(G.d_text, n)
        G.ix.set_reference(n, 11, G.pac)
    return G


def barrier(ctx):
    if ctx.dist is not None:
        ctx.dist.barrier()
    ctx.torch.cuda.synchronize()


def max_over_ranks(ctx, x):
    if ctx.dist is None:
        return x
    t = ctx.torch.tensor([x], device="cuda" if ctx.backend == "nccl" else "cpu", dtype=ctx.torch.float64)
    ctx.dist.all_reduce(t, op=ctx.dist.ReduceOp.MAX)
    return float(t.item())


def se_workload(ctx, G, kind, n_reads, steps, warmup, primary, cpu_seconds, want_e2e):
    """One single-end workload on genome G: kind = "headline" (configs[1]), "adna" (config 5's reads and options), "repeats" (the headline
    reads on the repeat-family genome).  primary: the run's main workload (PCIe-inclusive legs, --pipeline, the e2e leg)."""
    args, nabwa, synth, T, torch = ctx.args, ctx.nabwa, ctx.synth, ctx.T, ctx.torch
    rank, world, dev, ix, n = ctx.rank, ctx.world, ctx.dev, G.ix, G.n
    adna = kind == "adna"
    read_len, sub_ppm, indel_ppm = (76, 10000, args.indel_ppm) if adna else (args.read_len, args.sub_ppm, args.indel_ppm)
    # reads: this rank's shard (seeded by rank)
    seq, rseq, off = synth.synth_reads(G.d_text, n, n_reads, read_len, sub_ppm, indel_ppm, 2 + 1000 * rank, device=dev)
    opt = nabwa.gap_init_opt()
    if adna:
        seq, rseq, off = adna_profile(seq, n_reads, read_len, 5 + 1000 * rank)
        opt.fnr, opt.max_diff, opt.max_gapo, opt.seed_len = 0.01, -1, 2, 16500
    quick = quick_env()        # experiments on slow workloads: only the resident steps (no PCIe legs, no touch count)
    pipeline = args.pipeline and primary
    t_pcie = t_pcie_first = None
    n_pc = 0
    if primary:
        # one end-to-end pass over host buffers (upload + both kernels + compacted download): the PCIe-inclusive rate
        ix.cal_sa_reg_gap_flat(opt, seq[:off[1000]], rseq[:off[1000]], off[:1001], per_read=True)   # (loads the kernels' code objects once)
        torch.cuda.synchronize()
        n_pc = 1000 if quick else n_reads
        seq_pc, rseq_pc, off_pc = (seq[:off[n_pc]], rseq[:off[n_pc]], off[:n_pc + 1]) if quick else (seq, rseq, off)
        t_pcie_first = time.time()
        _na, _rows, _maxe = ix.cal_sa_reg_gap_flat(opt, seq_pc, rseq_pc, off_pc, per_read=True)      # the C one-shot entry on host buffers
        t_pcie_first = time.time() - t_pcie_first           # first call: the working buffers come from hipMalloc
        del _na, _rows, _maxe
        t_pcie = time.time()
        _na, _rows, _maxe = ix.cal_sa_reg_gap_flat(opt, seq_pc, rseq_pc, off_pc, per_read=True)      # what a streaming caller sees: buffers from the index's pool
        t_pcie = time.time() - t_pcie
        if rank == 0 and os.environ.get("NABWA_BENCH_MAXE"):          # how large the searches' stacks get (bwa_seq_t.max_entries)
            log("max_entries: percentiles 50/90/99/99.9/100 = %s; reads over 1024/4096/16384/65536/262144: %s"
                % (np.percentile(_maxe, [50, 90, 99, 99.9, 100]).tolist(), [int((_maxe > c).sum()) for c in (1024, 4096, 16384, 65536, 262144)]))
        del _na, _rows, _maxe
    batch = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)

    # Optional double buffering, as a streaming worker does: a second device-resident copy of the batch on its own stream.  Steps
    # alternate between the two, and a step is only waited for after the next one is enqueued, so the width kernel of
    # step k+1 runs in the straggler tail of the search kernel of step k (a launch cannot end before its longest search;
    # for most of that time most CUs are idle).  Every step is still one full pass (both kernels) over one 10 M-read batch.
    # Off by default: the headline number is steps strictly one after the other on one batch (--pipeline: +7 %).
    batches = [batch, nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)] if pipeline else [batch]
    n2 = 0
    kms, wms, dms = [], [], []
    for _ in range(warmup):
        for b in batches:
            b.run()
            n2 = b.sync()
        if pipeline:                                   # un-pipelined launches: the kernels' own durations
            kms.append(batch.last_kernel_ms())
            wms.append(batch.last_width_ms())
            dms.append(batch.last_deep_ms())
    barrier(ctx)
    t1 = time.time()
    for k in range(steps):
        batches[k % len(batches)].run()
        if not pipeline:
            n2 = batch.sync()
            kms.append(batch.last_kernel_ms())
            wms.append(batch.last_width_ms())
            dms.append(batch.last_deep_ms())
        elif k >= 1:
            n2 = max(n2, batches[(k - 1) % 2].sync())
    if pipeline:
        n2 = max(n2, batches[(steps - 1) % 2].sync())
        if not kms:
            kms, wms = [batches[(steps - 1) % 2].last_kernel_ms()], [batches[(steps - 1) % 2].last_width_ms()]
            dms = [batches[(steps - 1) % 2].last_deep_ms()]
    barrier(ctx)
    elapsed = max_over_ranks(ctx, time.time() - t1)
    checksum, n_rows = batch.checksum()
    # the rows of the TIMED run, for the comparison with the CPU (the untimed instrumented run below overwrites the batch's results;
    # until round 2 the comparison read those by mistake)
    timed_rows = batch.fetch_flat() if ctx.want_cpu else None

    e2e = None
    if want_e2e:
        e2e = e2e_leg(ctx, G, opt, seq, rseq, off, read_len)
    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (fm_search, first pass): algorithmic bytes / event time
        t_search, t_width = (0, 0) if quick else batch.count_touches()
        instrumented_same = True if quick else batch.checksum() == (checksum, n_rows)      # the untimed touch-counting run (tables and text mode off) must give the same rows
        if not instrumented_same:
            log("ERROR: the instrumented run's rows differ from the timed run's (checksum %016x / %d rows vs %016x / %d): no roofline figure is reported"
                % (batch.checksum() + (checksum, n_rows)))
        half_reads = (int(off[-1]) + n_reads) // 2
        s_ms, w_ms, d_ms = float(np.mean(kms)), float(np.mean(wms)), float(np.mean(dms)) if dms else 0.0
        # bwt_match_gap runs in two kernels: S (one read per lane) and, for the searches S hands on (arena outgrown / still running
        # after NABWA_TRIP_BUDGET trips), D (one read per wavefront).  Their event times add up to the search time of a pass.
        k_ms = s_ms + d_ms
        deep = d_ms > s_ms                      # deep searches (--adna): most of the work is kernel D's
        # dominant kernel = fm_search (bwt_match_gap): its own algorithmic bytes / its own event time
        bytes_alg = 48 * t_search + half_reads + 16 * n_rows
        bytes_w = 48 * t_width + half_reads
        achieved = bytes_alg / (k_ms * 1e-3) / 1e9
        full = n == GRCH38_LEN and read_len == (76 if adna else 100) and sub_ppm == (10000 if adna else 2000) and indel_ppm == 0
        traffic = pmc_traffic(kind, n_reads) if full else None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2) if instrumented_same else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5) if instrumented_same else None,
                    "traffic": traffic,
                    "kernel": ("fm_deep_kernel (one search per wavefront) + fm_search_kernel<false,false> before it" if deep else
                               "fm_search_kernel<false,false> (one read per lane) + fm_deep_kernel for the searches it hands on"),
                    "kernel_ms": round(k_ms, 3), "search_kernel_ms": round(s_ms, 3), "deep_kernel_ms": round(d_ms, 3),
                    "note": "achieved = the REFERENCE algorithm's bucket bytes / kernel time (an effective rate: the interval tables and text mode skip most of those touches); "
                            "traffic = HBM bytes per launch the counters saw in the committed pass of this kernel version (profiles/), null where none was taken",
                    "bytes_per_read": round(bytes_alg / n_reads, 1),
                    "bucket_touches_per_read": round(t_search / n_reads, 1),
                    "width_kernel": {"kernel": "fm_width_kernel<false>", "kernel_ms": round(w_ms, 3),
                                     "achieved": round(bytes_w / (w_ms * 1e-3) / 1e9, 2),
                                     "frac": round(bytes_w / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     "bucket_touches_per_read": round(t_width / n_reads, 1)},
                    "both_kernels": {"ms": round(k_ms + w_ms, 3),
                                     "achieved": round((bytes_alg + bytes_w) / ((k_ms + w_ms) * 1e-3) / 1e9, 2),
                                     "frac": round((bytes_alg + bytes_w) / ((k_ms + w_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
        cpu = None
        bit_exact = None
        if ctx.want_cpu:
            cpu, bit_exact = cpu_baseline(T, G.host_bwt, opt, seq, rseq, off, timed_rows, cpu_seconds if world == 1 else min(cpu_seconds, 3.0))
            if world > 1:
                cpu = None          # reported at N = 1 only: at N > 1 the host cores belong to the ranks (the sample check above stays)
        reads_per_s = n_reads * world * steps / elapsed
        what = "50-76 bp damaged SE, ancient-DNA options" if adna else "%d bp SE" % read_len
        out = {"metric": "aligned reads/s to GRCh38 (%s), FM-index search (bwa_cal_sa_reg_gap)%s%s" % (
                   what, ", bit-exact vs CPU" if bit_exact else (", CPU sample DIFFERS" if bit_exact is False else ", CPU comparison not run"),
                   "" if instrumented_same else ", INSTRUMENTED RUN DIFFERS (no roofline figure)"),
               "value": round(reads_per_s, 1), "unit": "reads/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": (("GRCh38-sized synthetic genome (%d bp) with repeat families (60 k LINE-like copies of a 6 kb consensus at 3-20%% divergence, 1 M Alu-like copies "
                                       "of a 300 bp consensus at 10-15%%, every tenth at 1-3%%, 200 k tandem repeats), " % n) if G.repeats else
                                      ("GRCh38-sized synthetic genome (%d bp, uniform ACGT + 2000 planted 5 kb repeats), " % n))
                                      + ("%d SE reads/GPU of 50-76 bases with terminal deamination + 1%% subs (SURVEY 8d C5), "
                                         "gap_opt_t of -n 0.01 -o 2 -l 16500, index replicated per GPU" % n_reads if adna else
                                         "%d x %d bp SE reads/GPU at %.1f%% subs, default gap_opt_t, index replicated per GPU"
                                         % (n_reads, read_len, sub_ppm / 1e4)),
                          "reads_per_gpu": n_reads, "read_len": read_len, "genome_len": n,
                          "parallelism": "reads sharded x%d, index replicated" % world,
                          "pipelining": "steps alternate between two device-resident batches on two streams" if pipeline else "none",
                          "single_batch_ms": round(k_ms + w_ms, 3),
                          "second_pass_reads": n2, "hits": n_rows, "hit_rows_per_read": round(n_rows / n_reads, 4), "checksum": "%016x" % checksum,
                          "bit_exact_vs_cpu_sample": bit_exact, "instrumented_run_same_rows": instrumented_same},
               "roofline": roofline, "cpu_baseline": cpu}
        if primary:
            out["config"]["pcie_inclusive_reads_per_s"] = round(n_pc / t_pcie, 1)
            out["config"]["pcie_inclusive_first_call_reads_per_s"] = round(n_pc / t_pcie_first, 1)
            out["e2e"] = e2e
    for b in batches:
        b.close()
    return out


def quick_env():
    return os.environ.get("NABWA_BENCH_QUICK") == "1"


def pack_text(d_text, n):
    """the synthetic genome as .pac bytes (4 bases per byte, first base in the top bits; reference bwtaln.h:33)"""
    t = d_text.to_host(np.uint8)[:n]
    pad = (-n) % 4
    if pad:
        t = np.concatenate([t, np.zeros(pad, np.uint8)])
    t = t.reshape(-1, 4)
    return np.ascontiguousarray((t[:, 0] << 6) | (t[:, 1] << 4) | (t[:, 2] << 2) | t[:, 3]).astype(np.uint8)


def ref_full_index(T, host_bwt, host_sa, pac, n):
    """the compiled reference's index structures around the arrays the GPU builder made (oracle/ref_harness.c)"""
    ref = T.load_ref()
    if ref is None:
        return None, None
    ref.ref_index_wrap_full.restype = C.c_void_p
    ref.ref_index_wrap_full.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_int]
    rix = C.c_void_p(ref.ref_index_wrap_full(T.ptr(host_bwt[0]), len(host_bwt[0]), T.ptr(host_bwt[1]), len(host_bwt[1]),
                                             T.ptr(host_sa[0]), T.ptr(host_sa[1]), T.ptr(pac), n, 11, 16))
    return ref, rix


def e2e_leg(ctx, G, opt, seq, rseq, off, L):
    """End to end on the first e2e_reads reads of the batch: unaligned BAM records in host memory -> the library's batch front-end
    (record parsing, tag erase, bam1_to_seq, FM search, posn_singleton on the drand48 stream, bwt_sa batch, refinement, MD/NM,
    bwa_update_bam1) -> aligned BAM records in host memory.  No BGZF on either side (the reference's bgzf.c is host I/O).
    Every rank runs its own batch (its shard's reads); the rate is all ranks' reads over the slowest rank's time.  On rank 0 a sample
    is compared, field by field incl. CIGAR / NM / MD, with the reference's own chain (bwa_cal_sa_reg_gap -> bwa_aln2seq_core ->
    bwa_cal_pac_pos_core -> bwa_refine_gapped) on the same reads, and that chain's time on the host threads is the leg's CPU baseline."""
    import struct
    args, nabwa, T, n = ctx.args, ctx.nabwa, ctx.T, G.n
    ix, pac = G.ix, G.pac
    n_e = min(args.e2e_reads, len(off) - 1)
    rec_len = 36 + 10 + (L + 1) // 2 + L
    rec = np.zeros((n_e, rec_len), np.uint8)
    rec[:, 0:4] = np.frombuffer(struct.pack("<I", rec_len - 4), np.uint8)
    core = struct.pack("<iiIIiiii", -1, -1, (4680 << 16) | 10, 4 << 16, L, -1, -1, 0)
    rec[:, 4:36] = np.frombuffer(core, np.uint8)
    names = np.char.zfill(np.arange(n_e).astype("U8"), 8)
    rec[:, 36] = ord("r")
    rec[:, 37:45] = np.frombuffer("".join(names).encode(), np.uint8).reshape(n_e, 8)
    fwd = seq[:n_e * L].reshape(n_e, L)[:, ::-1]                       # bwa_seq_t.seq is the read reversed
    code16 = np.array([1, 2, 4, 8, 15], np.uint8)[fwd]
    if L % 2:
        code16 = np.concatenate([code16, np.zeros((n_e, 1), np.uint8)], axis=1)
    rec[:, 46:46 + (L + 1) // 2] = (code16[:, 0::2] << 4) | code16[:, 1::2]
    rec[:, 46 + (L + 1) // 2:] = 40
    buf = np.ascontiguousarray(rec).reshape(-1)
    boff = np.arange(n_e + 1, dtype=np.int64) * rec_len
    Lb = nabwa.lib()
    P = C.c_void_p
    Lb.nabwa_isize_table_create.restype = P
    Lb.nabwa_isize_table_create.argtypes = [C.c_double, C.c_int64]
    Lb.nabwa_bam_batch_create.argtypes = [P, P, P, C.c_int, P, P, P]
    Lb.nabwa_bam_batch_pass1.argtypes = [P, P, P]
    Lb.nabwa_bam_batch_search.argtypes = [P]
    Lb.nabwa_bam_batch_pass2.argtypes = [P, P, P, P]
    Lb.nabwa_bam_batch_output.argtypes = [P, P, C.c_int64, P, P]
    Lb.nabwa_bam_batch_destroy.argtypes = [P]
    Lb.nabwa_isize_table_destroy.argtypes = [P]
    po = nabwa.pe_opt_default()

    def once(ob=None, oo=None):
        tab = P(Lb.nabwa_isize_table_create(po.ap_prior, n))
        st = C.c_uint64(nabwa.srand48_state(11))
        h = P()
        t = [time.time()]
        assert Lb.nabwa_bam_batch_create(ix._h, C.byref(opt), C.byref(po), n_e, T.ptr(buf), T.ptr(boff), C.byref(h)) == 0, Lb.nabwa_last_error()
        t.append(time.time())
        assert Lb.nabwa_bam_batch_pass1(h, C.byref(st), tab) == 0, Lb.nabwa_last_error()
        t.append(time.time())
        tot, mp = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        assert Lb.nabwa_bam_batch_pass2(h, tab, tot, mp) == 0, Lb.nabwa_last_error()
        t.append(time.time())
        nb = C.c_int64()
        if oo is None:
            oo = np.zeros(n_e + 1, np.int64)
        Lb.nabwa_bam_batch_output(h, None, 0, T.ptr(oo), C.byref(nb))
        if ob is None or len(ob) < nb.value:             # (a streaming caller keeps its output buffer from batch to batch)
            ob = np.zeros(max(nb.value, 1), np.uint8)
        assert Lb.nabwa_bam_batch_output(h, T.ptr(ob), nb.value, T.ptr(oo), C.byref(nb)) == 0
        t.append(time.time())
        Lb.nabwa_bam_batch_destroy(h)
        Lb.nabwa_isize_table_destroy(tab)
        return np.diff(t), ob, oo, nb

    dt_first, ob, oo, _ = once()            # the first batch of a size pays for its working buffers (device pool, host records, the caller's output buffer)
    barrier(ctx)
    t0 = time.time()
    dt, ob, oo, nb = once(ob, oo)           # what a streaming caller sees from then on
    barrier(ctx)
    elapsed = max_over_ranks(ctx, time.time() - t0)

    def streamed(n_batches):
        """the same batch n_batches times through the calls as a stream: create | search | pass 1 | pass 2 | output + destroy, a thread each (ctypes
        releases the interpreter lock), batches in order through every stage -- pass 1 on the one drand48 stream in input order, as nabwa_bam2bam
        runs it; for single-end records pass 2 needs nothing of later batches, so it may run beside the next batch's pass 1.  While a batch's
        search runs on the GPU the host threads work on its neighbours."""
        import queue
        import threading
        q0, q1, q2, q3 = queue.Queue(1), queue.Queue(1), queue.Queue(1), queue.Queue(1)      # (one batch waiting between two stages: what is in flight stays within the pool of per-batch blocks)
        tab = P(Lb.nabwa_isize_table_create(po.ap_prior, n))
        st = C.c_uint64(nabwa.srand48_state(11))
        outs = [(np.zeros(len(ob), np.uint8), np.zeros(n_e + 1, np.int64)) for _ in range(2)]
        for o in outs:
            o[0][:] = 1                     # (touched: a streaming caller's buffers are)
        err = []

        def s_create():
            for _ in range(n_batches):
                h = P()
                if Lb.nabwa_bam_batch_create(ix._h, C.byref(opt), C.byref(po), n_e, T.ptr(buf), T.ptr(boff), C.byref(h)) != 0:
                    err.append(Lb.nabwa_last_error())
                    h = None
                q0.put(h)
                if h is None:
                    return
            q0.put(None)

        def s_search():                     # the GPU's part of pass 1 (nabwa_bam_batch_search: upload, kernels W / S / D, rows back), as nabwa_bam2bam's search threads run it
            while True:
                h = q0.get()
                if h is not None and Lb.nabwa_bam_batch_search(h) != 0:
                    err.append(Lb.nabwa_last_error())
                q1.put(h)
                if h is None:
                    return

        def s_pass1():
            while True:
                h = q1.get()
                st.value = nabwa.srand48_state(11)          # (every batch of the stream is the same batch: the same draws, the same bytes -- checked below)
                if h is not None and Lb.nabwa_bam_batch_pass1(h, C.byref(st), tab) != 0:
                    err.append(Lb.nabwa_last_error())
                q2.put(h)
                if h is None:
                    return

        def s_pass2():
            while True:
                h = q2.get()
                if h is not None:
                    tot, mp = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
                    if Lb.nabwa_bam_batch_pass2(h, tab, tot, mp) != 0:
                        err.append(Lb.nabwa_last_error())
                q3.put(h)
                if h is None:
                    return

        def s_out():
            k = 0
            while True:
                h = q3.get()
                if h is None:
                    return
                o_b, o_o = outs[k & 1]
                nbk = C.c_int64()
                if Lb.nabwa_bam_batch_output(h, T.ptr(o_b), len(o_b), T.ptr(o_o), C.byref(nbk)) != 0:
                    err.append(Lb.nabwa_last_error())
                Lb.nabwa_bam_batch_destroy(h)
                k += 1

        th = [threading.Thread(target=f) for f in (s_create, s_search, s_pass1, s_pass2, s_out)]
        t_s = time.time()
        for x in th:
            x.start()
        for x in th:
            x.join()
        t_s = time.time() - t_s
        Lb.nabwa_isize_table_destroy(tab)
        assert not err, err[:2]
        return t_s, outs

    n_stream = int(os.environ.get("NABWA_BENCH_E2E_BATCHES", "8"))
    streamed(2)                                 # (the pool of per-batch blocks grows to what four batches in flight need)
    barrier(ctx)
    t_stream, outs = streamed(n_stream)
    barrier(ctx)
    t_stream = max_over_ranks(ctx, t_stream)
    stream_same = bool(np.array_equal(outs[(n_stream - 1) & 1][0][:nb.value], ob[:nb.value]))      # the last batch of the stream = the one-batch run's bytes
    if ctx.rank != 0:
        return None
    exact, n_chk, cpu = None, 0, None
    if ctx.want_cpu:
        ref, rix = ref_full_index(T, G.host_bwt, G.host_sa, pac, n)
        if ref is not None:
            import bamlib
            cores = cpu_threads()
            n_chk = min(n_e, int(os.environ.get("NABWA_BENCH_E2E_SAMPLE", "100000")))
            copt = T.GapOpt()
            C.memmove(C.byref(copt), C.byref(opt), 64)
            ref.ref_cal_sa_reg_gap_mt.restype = C.c_long
            ref.ref_cal_sa_reg_gap_mt.argtypes = [P, P, C.c_int, P, P, P, C.c_int, P, P, C.c_long]
            ref.ref_se_chain_mt.argtypes = [P, P, C.c_int, C.c_int, P, P, P, P, P, C.c_int, P, P, P, C.c_int, P]
            na = np.zeros(n_chk, np.int32)
            rows = np.zeros(64 * n_chk + 4096, T.ALN_DT)
            o = np.ascontiguousarray(off[:n_chk + 1])
            t_s = time.time()
            assert ref.ref_cal_sa_reg_gap_mt(rix, C.byref(copt), n_chk, T.ptr(o), T.ptr(seq), T.ptr(rseq), cores, T.ptr(na), T.ptr(rows), len(rows)) >= 0
            t_s = time.time() - t_s
            f = np.zeros((n_chk, 16), np.int64)
            cg = np.zeros((n_chk, 64), np.uint16)
            MDC = 256
            md = np.zeros((n_chk, MDC), np.uint8)
            secs = (C.c_double * 2)()
            ref.ref_seed48(11)
            ref.ref_se_chain_mt(rix, C.byref(copt), po.max_occ_se, n_chk, T.ptr(o), T.ptr(seq), T.ptr(rseq), T.ptr(na), T.ptr(rows), cores,
                                T.ptr(f), T.ptr(cg), T.ptr(md), MDC, secs)
            dec = bamlib.decode(ob, oo[:n_chk + 1], ["synth%d" % (k + 1) for k in range(16)])
            exact = True
            n_bad = 0
            for i in range(n_chk):
                g, fi = dec[i], f[i]
                if fi[0] == 0:
                    ok = bool(g["flag"] & 4) and g["rname"] == "*"
                else:
                    tg = g["tags"]
                    cid = int(g["rname"][5:]) - 1
                    bridging = bool(g["flag"] & 4)
                    ncg = int(fi[12])
                    cig = "".join("%d%s" % (c & 0x3fff, "MIDS"[c >> 14]) for c in cg[i, :ncg]) if ncg else "%dM" % fi[14]
                    mds = bytes(md[i]).split(b"\0", 1)[0].decode()
                    ok = (g["pos"] + n * cid // 16 == fi[9] + 1 and bool(g["flag"] & 16) == bool(fi[1]) and (g["mapq"] == fi[10] or bridging)
                          and tg["X0"] == fi[7] and tg.get("X1", fi[8]) == fi[8] and tg["XM"] == fi[2] and tg["XO"] == fi[3] and tg["XG"] == fi[3] + fi[4]
                          and (bridging or (g["cigar"] == cig and tg["NM"] == fi[13] and tg["MD"] == mds and tg["XT"] == " URM"[fi[0]])))
                if not ok:
                    n_bad += 1
                    if n_bad <= 3:
                        log("e2e sample: record %d differs: %s vs reference %s" % (i, g, fi.tolist()))
                exact = exact and bool(ok)
            t_cpu = t_s + secs[0] + secs[1]
            cpu = {"value": round(n_chk / t_cpu, 1), "unit": "reads/s", "cores": cores, "kind": "reference",
                   "sample": "first %d reads of the same batch through the reference's own functions: bwa_cal_sa_reg_gap on %d threads (%.2f s), posn_singleton serial "
                             "on the one drand48 stream (%.2f s), bwa_refine_gapped on %d threads (%.2f s); BAM parsing and bwa_update_bam1 not included (bam2bam.c needs <zmq.h>)"
                             % (n_chk, cores, t_s, secs[0], cores, secs[1])}
            if ctx.world > 1:
                cpu = None
    return {"what": "unaligned BAM records in host memory -> aligned BAM records in host memory (nabwa_bam_batch_*: the whole of bam2bam's two passes for single-end records, without BGZF)",
            "reads": n_e, "n_gpus": ctx.world, "reads_per_s": round(n_e * ctx.world / elapsed, 1), "first_batch_reads_per_s": round(n_e / dt_first.sum(), 1), "bam_bytes_out": int(nb.value),
            "streamed_reads_per_s": round(n_e * n_stream * ctx.world / t_stream, 1),
            "streamed": "%d batches of %d records through create | search | pass 1 | pass 2 | output + destroy as a five-thread pipeline over the same calls (what nabwa_bam2bam does around them, "
                        "without BGZF); reads_per_s is ONE batch through the four calls one after the other; the stream's last batch has the one-batch run's bytes: %s" % (n_stream, n_e, stream_same),
            "stage_ms": {"parse + erase tags + bam1_to_seq": round(dt[0] * 1e3, 1), "pass 1: search (upload, kernels W / S / D, rows back) + posn_singleton": round(dt[1] * 1e3, 1),
                         "pass 2: bwa_refine_gapped + MD/NM + bwa_update_bam1": round(dt[2] * 1e3, 1), "records out": round(dt[3] * 1e3, 1)},
            "bit_exact_vs_reference_sample": exact, "sample_reads": n_chk,
            "sample_fields": "flag, contig, position, strand, MAPQ, CIGAR, NM, MD, XT, X0, X1, XM, XO, XG", "cpu_baseline": cpu}


def cpu_threads():
    # the GPU box gives one GPU's job a share of 16 host threads (pool-sizing rule of the box); NABWA_BENCH_CPU_THREADS overrides
    return min(len(os.sched_getaffinity(0)), int(os.environ.get("NABWA_BENCH_CPU_THREADS", "16")))


def pe_workload(ctx, G, N, steps, warmup, cpu_seconds):
    """BASELINE config 3: the paired-end path.  One step = both ends searched (kernels W / S / D on 2 N reads, resident), the
    rows fetched, posn_pair on the host's drand48 stream + bwt_sa batch, the insert-size estimate from the histogram
    (infer_isize_hist), finish_pair (pairing, mate rescue and gap refinement as GPU batches).  value = pairs/s."""
    nabwa, synth, T = ctx.nabwa, ctx.synth, ctx.T
    rank, world, dev, ix, n = ctx.rank, ctx.world, ctx.dev, G.ix, G.n
    L = 150
    seq, rseq, off = synth.synth_pairs(G.d_text, n, N, L, 20000, 100000, 400.0, 40.0, 3 + 1000 * rank, device=dev)
    opt = nabwa.gap_init_opt()
    po = nabwa.pe_opt_default()
    full = np.full(2 * N, L, np.int32)
    batch = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
    rec_buf = (nabwa.PeRec * (2 * N))()          # the per-end records (3 KB each), allocated once as a streaming caller would
    keep = {}

    def step():
        t = [time.time()]
        batch.run()
        n2 = batch.sync()
        t.append(time.time())
        n_aln, rows, _ = batch.fetch_flat(keep)          # (the caller's arrays from step to step, as its records are)
        t.append(time.time())
        recs, _ = ix.pe_posn_flat(opt, off, full, n_aln, rows, nabwa.srand48_state(11), out=rec_buf)
        t.append(time.time())
        h = np.zeros(100000, np.uint16)
        nabwa.isize_add_pairs(recs, N, h)                # improve_isize_est (insert_size.c:141-165)
        rc, ii = nabwa.isize_infer(h, po.ap_prior, n)
        t.append(time.time())
        tot, mp = ix.pe_finish_flat(opt, po, ii, seq, rseq, off, n_aln, rows, recs)
        t.append(time.time())
        return recs, ii, n2, (n_aln, rows), np.diff(t), (tot, mp), (batch.last_kernel_ms(), batch.last_width_ms(), batch.last_deep_ms())

    for _ in range(warmup):
        step()
    barrier(ctx)
    t1 = time.time()
    splits, kms = [], []
    for _ in range(steps):
        recs, ii, n2, hits, dt, sw, km = step()
        splits.append(dt)
        kms.append(km)
    barrier(ctx)
    elapsed = max_over_ranks(ctx, time.time() - t1)
    # the same steps as a stream: the next batch's search runs on the GPU (nabwa_batch_run returns at once) while the host finishes this one -- two
    # batch objects over the same resident reads, taken in turn; the results of every step are the sequential step's
    streamed = None
    if not quick_env() or os.environ.get("NABWA_BENCH_STREAM"):
        batch2 = nabwa.Batch(ix, opt, seq, rseq, off, per_read=True)
        rec_buf2 = (nabwa.PeRec * (2 * N))()
        pair = [(batch, rec_buf, {}), (batch2, rec_buf2, {})]

        import queue
        import threading
        n_st = max(steps, 4)
        qs, qf = queue.Queue(1), queue.Queue(2)
        for k in range(2):
            qf.put(k)

        def searcher():                                # run + sync + fetch of batch k & 1, one after the other: the GPU's share of a step
            for k in range(n_st + 1):
                slot = qf.get()
                bk, rb, kp = pair[slot]
                bk.run()
                bk.sync()
                na, rw, _ = bk.fetch_flat(kp)
                qs.put((slot, na, rw))

        def finish():                                  # the host's share, while the searcher is at the next batch
            slot, na, rw = qs.get()
            rb = pair[slot][1]
            rc_, _ = ix.pe_posn_flat(opt, off, full, na, rw, nabwa.srand48_state(11), out=rb)
            hh = np.zeros(100000, np.uint16)
            nabwa.isize_add_pairs(rc_, N, hh)
            _, ii_ = nabwa.isize_infer(hh, po.ap_prior, n)
            ix.pe_finish_flat(opt, po, ii_, seq, rseq, off, na, rw, rc_)
            qf.put(slot)
            return rc_

        th = threading.Thread(target=searcher)
        th.start()
        finish()                                       # (warm-up: the second batch's buffers)
        barrier(ctx)
        ts = time.time()
        for k in range(n_st):
            rc_last = finish()
        barrier(ctx)
        ts = max_over_ranks(ctx, time.time() - ts)
        th.join()
        same = bool(np.array_equal(np.frombuffer(rc_last, np.uint8).reshape(2 * N, -1)[:, :64], np.frombuffer(recs, np.uint8).reshape(2 * N, -1)[:, :64]))
        streamed = {"pairs_per_s": round(N * world * n_st / ts, 1), "ms_per_step": round(ts / n_st * 1e3, 1), "steps": n_st,
                    "what": "the same step with the next batch's search on the GPU while the host positions and finishes this one (two batch objects over the resident reads); "
                            "the heads of the finished records equal the sequential step's: %s" % same}
        batch2.close()
        del rec_buf2
    if rank != 0:
        batch.close()
        return None
    sp = np.mean(splits, axis=0) * 1e3
    checksum, n_rows = batch.checksum()
    t_search, t_width = (0, 0) if quick_env() else batch.count_touches()
    instrumented_same = True if quick_env() else batch.checksum() == (checksum, n_rows)
    s_ms, w_ms, d_ms = [float(x) for x in np.mean(kms, axis=0)]
    bytes_alg = 48 * t_search + N * L + 16 * n_rows
    k_ms = s_ms + d_ms
    roofline = {"bound": "hbm", "achieved": round(bytes_alg / (k_ms * 1e-3) / 1e9, 2) if instrumented_same else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(bytes_alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if instrumented_same else None,
                "traffic": pmc_traffic("pe", N) if n == GRCH38_LEN else None,
                "kernel": "fm_search_kernel<false,false> + fm_deep_kernel (bwt_match_gap of both ends)", "kernel_ms": round(k_ms, 3),
                "search_kernel_ms": round(s_ms, 3), "deep_kernel_ms": round(d_ms, 3), "width_kernel_ms": round(w_ms, 3),
                "bytes_per_read": round(bytes_alg / (2 * N), 1), "bucket_touches_per_read": round(t_search / (2 * N), 1),
                "note": "the FM search is the dominant GPU kernel of the step; the finishing chain is host-bound (stage_ms)"}
    cpu, exact = None, None
    if ctx.want_cpu:
        cpu, exact = pe_cpu_baseline(N, cpu_seconds if world == 1 else min(cpu_seconds, 3.0), nabwa, T, G.host_bwt, G.host_sa, G.pac, n, opt, seq, rseq, off, hits, recs, ii, L)
        if world > 1:
            cpu = None
    tp = np.frombuffer(recs, np.uint8).reshape(2 * N, C.sizeof(nabwa.PeRec))[:, nabwa.PeRec.se.offset + nabwa.SeRec.type.offset]
    out = {"metric": "aligned pairs/s to GRCh38 (2 x 150 bp PE at 2%% error), search + posn_pair + insert-size estimate + finish_pair%s%s"
                     % (", bit-exact vs CPU" if exact else (", CPU sample DIFFERS" if exact is False else ", CPU comparison not run"),
                        "" if instrumented_same else ", INSTRUMENTED RUN DIFFERS (no roofline figure)"),
           "value": round(N * world * steps / elapsed, 1), "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic",
           "config": {"workload": "GRCh38-sized synthetic genome (%d bp), %d pairs/GPU of 2 x %d bp, 2%% substitutions, 10%% of the reads with a 1-base indel, "
                                  "inserts ~ N(400, 40), default gap_opt_t / pe_opt_t (BASELINE config 3 at %d of its 10 M pairs per step)" % (n, N, L, N),
                      "pairs_per_gpu": N, "read_len": L, "genome_len": n, "parallelism": "pairs sharded x%d, index replicated" % world,
                      "stage_ms": {"search (kernels W, S, D)": round(float(sp[0]), 1), "rows to the host": round(float(sp[1]), 1),
                                   "posn_pair (host RNG + bwt_sa batch)": round(float(sp[2]), 1), "insert-size estimate": round(float(sp[3]), 1),
                                   "finish_pair (pairing, mate rescue, refinement, MD)": round(float(sp[4]), 1)},
                      "isize": [ii.avg, ii.std, ii.low, ii.high, ii.high_bayesian], "mate_rescued": int(sw[1][0]), "rescue_attempts": int(sw[0][0]),
                      "mapped_ends": int((tp != 0).sum()), "second_pass_reads": n2, "hits": n_rows, "checksum": "%016x" % checksum,
                      "bit_exact_vs_cpu_sample": exact, "instrumented_run_same_rows": instrumented_same, "streamed": streamed},
           "roofline": roofline, "cpu_baseline": cpu}
    batch.close()
    del rec_buf
    return out


def pe_cpu_baseline(n_pairs, cpu_seconds, nabwa, T, host_bwt, host_sa, pac, n, opt, seq, rseq, off, hits, recs, ii, L):
    """the reference's own functions on a bounded sample of the same pairs: bwa_cal_sa_reg_gap of both ends on all host threads
    (ref_cal_sa_reg_gap_mt), then posn_pair serially and finish_pair on all threads (ref_pe_chain_mt); the GPU records of the
    sample are compared field for field"""
    cores = cpu_threads()
    ref, rix = ref_full_index(T, host_bwt, host_sa, pac, n)
    if ref is None:
        return None, None
    copt = T.GapOpt()
    C.memmove(C.byref(copt), C.byref(opt), 64)
    ref.ref_cal_sa_reg_gap_mt.restype = C.c_long
    ref.ref_cal_sa_reg_gap_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    ref.ref_pe_new.restype = C.c_void_p
    ref.ref_pe_new.argtypes = [C.c_int]
    ref.ref_pe_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ref.ref_pe_chain_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ref.ref_pe_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
    ref.ref_pe_free.argtypes = [C.c_void_p]

    def search(n_s):
        o = np.ascontiguousarray(off[:2 * n_s + 1])
        na = np.zeros(2 * n_s, np.int32)
        cap = 128 * n_s + 4096
        rows = np.zeros(cap, T.ALN_DT)
        t = time.time()
        tot = ref.ref_cal_sa_reg_gap_mt(rix, C.byref(copt), 2 * n_s, T.ptr(o), T.ptr(seq), T.ptr(rseq), cores, T.ptr(na), T.ptr(rows), cap)
        assert tot >= 0
        return time.time() - t, na, rows[:tot]

    pilot = 64 * cores
    dt, _, _ = search(pilot)
    n_s = int(min(n_pairs, max(pilot, pilot / max(dt, 1e-3) * cpu_seconds * 0.7)))
    t_search, na, rows = search(n_s)
    g_na, g_rows = hits
    b0 = np.concatenate([[0], np.cumsum(na)])
    gb = np.concatenate([[0], np.cumsum(g_na[:2 * n_s])])
    exact = bool(np.array_equal(na, g_na[:2 * n_s]) and rows.tobytes() == g_rows[:gb[-1]].tobytes())
    b = C.c_void_p(ref.ref_pe_new(n_s))
    for i in range(2 * n_s):
        r = np.ascontiguousarray(rows[b0[i]:b0[i + 1]])
        ref.ref_pe_set(b, i // 2, i % 2, L, T.ptr(np.ascontiguousarray(seq[off[i]:off[i + 1]])), T.ptr(np.ascontiguousarray(rseq[off[i]:off[i + 1]])), len(r), T.ptr(r))
    ref.ref_seed48(11)
    iiv = (C.c_double * 6)(ii.avg, ii.std, ii.ap_prior, ii.low, ii.high, ii.high_bayesian)
    secs = (C.c_double * 2)()
    ref.ref_pe_chain_mt(b, rix, C.byref(copt), iiv, cores, secs)
    f = np.zeros(17, np.int64); cg = np.zeros(256, np.uint16); mdb = C.create_string_buffer(1024); mu = np.zeros(21 * 16, np.int64)
    for i in range(2 * n_s):
        ref.ref_pe_get(b, i // 2, i % 2, T.ptr(f), T.ptr(cg), mdb, 1024, T.ptr(mu))
        g = recs[i]; s = g.se
        if f[0] == 0:
            exact = exact and s.type == 0
            continue
        bridging = bool(s.flag & 4)
        got = [s.type, s.strand, s.n_mm, s.n_gapo, s.n_gape, s.score, s.sa, s.c1, s.c2, s.pos, s.mapQ if not bridging else int(f[10]), s.seQ]
        same = got == [int(x) for x in f[:12]] and s.n_cigar == f[13] and list(s.cigar[:s.n_cigar]) == list(cg[:s.n_cigar]) and s.nm == f[14] \
            and s.md == mdb.value and s.n_multi == f[15] and (g.extra_flag & 0xff) == (f[12] & 0xff)
        exact = exact and bool(same)
    ref.ref_pe_free(b)
    t_all = t_search + secs[0] + secs[1]
    log("cpu baseline (reference): %d pairs on %d threads: search %.2f s, posn_pair %.2f s (serial, as in the reference), finish_pair %.2f s" % (n_s, cores, t_search, secs[0], secs[1]))
    return {"value": round(n_s / t_all, 1), "unit": "pairs/s", "cores": cores, "kind": "reference",
            "sample": "first %d pairs of the same batch: bwa_cal_sa_reg_gap of both ends on %d threads (%.1f s), posn_pair serial (%.1f s), finish_pair on %d threads (%.1f s)"
                      % (n_s, cores, t_search, secs[0], cores, secs[1])}, exact


def adna_profile(seq, n_reads, L, seed):
    """SURVEY 8d C5 on top of fixed-length synthetic reads: keep the first U{50..L} bases of every read and deaminate its
    ends (5' C>T, 3' G>A with probability 0.3 * 0.7^distance).  seq holds the reads REVERSED (bwa_seq_t.seq); returns
    (seq, rseq, off) of the new reads in the same encoding."""
    rng = np.random.default_rng(seed)
    out_s, lens_all = [], []
    for lo in range(0, n_reads, 1 << 20):
        hi = min(n_reads, lo + (1 << 20))
        fwd = seq[lo * L:hi * L].reshape(hi - lo, L)[:, ::-1].copy()
        m = hi - lo
        lens = rng.integers(50, L + 1, m).astype(np.int32)
        rows = np.arange(m)
        for j in range(12):
            pr = 0.3 * 0.7 ** j
            hit = (rng.random(m) < pr) & (fwd[:, j] == 1)
            fwd[hit, j] = 3                                       # C > T at the 5' end
            col = lens - 1 - j
            hit = (rng.random(m) < pr) & (fwd[rows, col] == 2)
            fwd[rows[hit], col[hit]] = 0                           # G > A at the 3' end
        k = np.arange(L, dtype=np.int32)[None, :]
        src = np.clip(lens[:, None] - 1 - k, 0, L - 1)            # reversed again: position k of seq = base len-1-k of the read
        rev = np.take_along_axis(fwd, src, axis=1)
        out_s.append(rev[k < lens[:, None]])
        lens_all.append(lens)
    s = np.concatenate(out_s)
    lens = np.concatenate(lens_all)
    off = np.zeros(n_reads + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    r = np.where(s < 4, 3 - s, s).astype(np.uint8)
    return np.ascontiguousarray(s), np.ascontiguousarray(r), off


def pmc_traffic(kind, units):
    """HBM bytes of one launch of the workload's dominant kernel from the committed counter pass of this kernel version (profiles/:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, profiles/collect_pmc*.sh; both are 64-byte fabric requests
    for these kernels' 64-byte gathers and 16-byte stores, so no gfx950 half-rate correction applies).  None when there is no pass
    for this workload at this size -- counters cannot be collected from inside the timed run."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    if any(k in os.environ for k in ("NABWA_KMER_T", "NABWA_TEXT_MODE", "NABWA_TRIP_BUDGET")):
        return None                                          # not the configuration the counter pass was taken on
    for name in PMC_FILES.get(kind, ()):
        try:
            d = json.load(open(os.path.join(here, name)))
            if int(d.get("units", units)) != int(units):
                continue
            k = d[d.get("dominant", "D" if kind != "headline" else "S")]     # the dominant kernel: S on the headline workload, D on the deep ones
            return round((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0, 0)
        except Exception:
            continue
    return None


# newest first; a file may name the launch size it was taken at ("units") and its dominant kernel ("dominant")
PMC_FILES = {"headline": ("r03_pmc.json", "r02_pmc.json"), "adna": ("r03_pmc_adna.json",), "pe": ("r03_pmc_pe.json",), "repeats": ("r03_pmc_repeats.json",)}


def cpu_baseline(T, host_bwt, opt, seq, rseq, off, timed_rows, cpu_seconds):
    """The reference's own bwa_cal_sa_reg_gap (oracle/_ref, compiled from /root/reference) on all host
    cores over a bounded sample of the same reads; falls back to the CPU restatement ("port") when the
    compiled reference did not travel.  Also checks the GPU rows of the sample bit-for-bit."""
    cores = cpu_threads()
    ref = T.load_ref()
    n_all = len(off) - 1
    got, _ = None, None

    def run(n_s, threads):
        o = np.ascontiguousarray(off[:n_s + 1])
        n_aln = np.zeros(n_s, np.int32)
        cap = 64 * n_s + 4096
        rows = np.zeros(cap, T.ALN_DT)
        t = time.time()
        if ref is not None:
            tot = ref.ref_cal_sa_reg_gap_mt(rix, C.byref(copt), n_s, T.ptr(o), T.ptr(seq), T.ptr(rseq), threads,
                                            T.ptr(n_aln), T.ptr(rows), cap)
        else:
            maxe = np.zeros(n_s, np.int32)
            tot = olib.orc_cal_sa_reg_gap(oix, C.byref(copt), n_s, T.ptr(o), T.ptr(seq), T.ptr(rseq), 1, T.ptr(n_aln),
                                          T.ptr(rows), cap, T.ptr(maxe), threads, None)
        dt = time.time() - t
        assert tot >= 0
        return dt, n_aln, rows[:tot]

    copt = T.GapOpt()
    C.memmove(C.byref(copt), C.byref(opt), 64)
    if ref is not None:
        ref.ref_index_wrap.restype = C.c_void_p
        ref.ref_index_wrap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        ref.ref_cal_sa_reg_gap_mt.restype = C.c_long
        ref.ref_cal_sa_reg_gap_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_long]
        rix = C.c_void_p(ref.ref_index_wrap(T.ptr(host_bwt[0]), len(host_bwt[0]), T.ptr(host_bwt[1]), len(host_bwt[1])))
        kind = "reference"
    else:
        olib = T.load_oracle()
        oix = C.c_void_p(olib.orc_index_wrap(T.ptr(host_bwt[0]), len(host_bwt[0]), T.ptr(host_bwt[1]), len(host_bwt[1])))
        kind = "port"
    pilot = min(n_all, 256 * cores)
    dt, _, _ = run(pilot, cores)
    n_s = int(min(n_all, max(pilot, pilot / max(dt, 1e-3) * cpu_seconds)))
    dt, n_aln, rows = run(n_s, cores)
    log("cpu baseline (%s): %d reads on %d threads in %.2f s" % (kind, n_s, cores, dt))
    # parity of the GPU rows (of the timed run) for the sample
    g_na, g_rows, _ = timed_rows
    gb = int(np.sum(g_na[:n_s]))
    exact = bool(np.array_equal(g_na[:n_s], n_aln) and g_rows[:gb].tobytes() == rows.tobytes())
    cpu = {"value": round(n_s / dt, 1), "unit": "reads/s", "cores": cores, "kind": kind,
           "sample": "first %d reads of the same batch, bwa_cal_sa_reg_gap once per read on %d host threads, %.1f s"
                     % (n_s, cores, dt)}
    return cpu, bool(exact)


if __name__ == "__main__":
    main()
