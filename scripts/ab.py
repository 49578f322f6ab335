#!/usr/bin/env python3
"""The one A/B driver (GPU box, through gpurun): parity first, then interleaved timing, then wave statistics.

usage: scripts/ab.py OUTDIR [--reps 3] [--iters 10] [--kinds "fwd prelim test"] [--pads "0"] [--parity] [--bench]
                     [--wstats] variant...

  variant   `main` = the in-tree libhf.so, NAME = scratch_so/libhf_NAME.so (scripts/vb.sh / variant_build.sh)
  --parity  run the GPU parity suite on every variant FIRST; a variant whose suite does not pass is reported as
            FAILED and is NOT timed (round 3 timed a library whose parity run had aborted; it then faulted the GPU)
  --pads    row pitches to time every variant at (prof_kernels.py --pad, floats between the SoA rows)
  --bench   additionally two bench.py runs (20 steps) per variant
  --wstats  wave statistics of NAME_ws / NAME_ws3 / NAME_ws4 builds where they exist (scripts/wstats.py)

Timing = prof_kernels.py per (variant, pad), interleaved over `reps` repetitions; the table is median [min-max] per
kind, written to gpurun_out/OUTDIR/table.txt next to all.log.  Every child runs under `timeout -k 10`.
"""
import argparse, collections, json, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARITY = ["tests/test_gpu_parity.py", "tests/test_gpu_full_size.py", "tests/test_gpu_sheared_stress.py", "tests/test_gpu_band.py",
          "tests/test_gpu_known_answers.py"]


def lib_of(v):
    return None if v == "main" else os.path.join(ROOT, "scratch_so", f"libhf_{v}.so")


def env_for(v):
    e = dict(os.environ)
    e.pop("HF_LIB", None)
    if lib_of(v):
        e["HF_LIB"] = lib_of(v)
    return e


def run(cmd, env, log, limit):
    """returns the exit code; output appended to `log`"""
    with open(log, "a") as f:
        f.write(f"$ {' '.join(cmd)}\n"); f.flush()
        return subprocess.call(["timeout", "-k", "10", str(limit)] + cmd, env=env, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out"); ap.add_argument("variants", nargs="+")
    ap.add_argument("--reps", type=int, default=3); ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--kinds", default="fwd prelim test sec_fwd"); ap.add_argument("--pads", default="0")
    ap.add_argument("--parity", action="store_true"); ap.add_argument("--bench", action="store_true")
    ap.add_argument("--wstats", action="store_true")
    ap.add_argument("--parity-tests", default=" ".join(PARITY))
    a = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out", a.out)
    os.makedirs(out, exist_ok=True)
    for v in a.variants:
        if lib_of(v) and not os.path.exists(lib_of(v)):
            sys.exit(f"{lib_of(v)} is missing")
    ok = list(a.variants)
    if a.parity:
        for v in a.variants:
            if v == "main":
                continue
            rc = run([sys.executable, "-m", "pytest"] + a.parity_tests.split() + ["-x", "-q", "-m", "gpu"], env_for(v),
                     os.path.join(out, f"parity_{v}.log"), 600)
            print(f"parity {v}: {'ok' if rc == 0 else 'FAILED rc=%d -- not timed' % rc}", flush=True)
            if rc != 0:
                ok.remove(v)
                if rc in (124, 137, -6, 134, -11, 139):   # timeout / abort / fault: no further GPU step in this call
                    sys.exit(f"variant {v} aborted the parity suite (rc {rc}): stopping")
    pads = [int(p) for p in a.pads.split()]
    allog = os.path.join(out, "all.log")
    d = collections.defaultdict(list)
    for rep in range(a.reps):
        for v in ok:
            for pad in pads:
                tag = v if len(pads) == 1 else f"{v}@{pad}"
                tmp = os.path.join(out, "_last.log")
                open(tmp, "w").close()
                rc = run([sys.executable, "scripts/prof_kernels.py", "--iters", str(a.iters), "--pad", str(pad)] + a.kinds.split(),
                         env_for(v), tmp, 300)
                with open(allog, "a") as f:
                    for l in open(tmp):
                        p = l.split()
                        if len(p) >= 4 and p[2] == "ms":
                            d[(tag, p[0])].append(float(p[1])); f.write(f"{tag} {l}")
                if rc != 0:
                    sys.exit(f"timing of {tag} failed (rc {rc}, see {tmp}): stopping")
    tags, kinds = [], []
    for (t, k) in d:
        if t not in tags: tags.append(t)
        if k not in kinds: kinds.append(k)
    lines = ["variant".ljust(14) + "".join(k.rjust(24) for k in kinds)]
    for t in tags:
        lines.append(t.ljust(14) + "".join(
            (f"{statistics.median(d[(t, k)]):.3f} [{min(d[(t, k)]):.3f}-{max(d[(t, k)]):.3f}]" if d[(t, k)] else "-").rjust(24) for k in kinds))
    table = "\n".join(lines)
    print(table, flush=True)
    open(os.path.join(out, "table.txt"), "w").write(table + "\n")
    if a.bench:
        for rep in range(2):
            for v in ok:
                e = env_for(v); e["HF_BENCH_EXTRAS"] = "0"
                p = subprocess.run(["timeout", "-k", "10", "300", sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--cpu-seconds", "0"],
                                   env=e, cwd=ROOT, capture_output=True, text=True)
                js = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
                if p.returncode != 0 or not js:
                    sys.exit(f"bench of {v} failed: {p.stderr[-400:]}")
                open(os.path.join(out, f"bench_{v}_{rep}.json"), "w").write(js[-1] + "\n")
                r = json.loads(js[-1])
                print(f"bench {v} #{rep}: {r['value']} Mrays/s, {r['ms_per_step']} ms/step, fwd {r['roofline']['fwd_ms']} adj {r['roofline']['adj_ms']}", flush=True)
    if a.wstats:
        for v in ok:
            for suf, mode in (("_ws", "1"), ("_ws3", "3"), ("_ws4", "4")):
                lib = os.path.join(ROOT, "scratch_so", f"libhf_{v}{suf}.so") if v != "main" else None
                if lib and os.path.exists(lib):
                    e = dict(os.environ); e["HF_LIB"] = lib
                    log = os.path.join(out, f"wstats{mode}_{v}.txt")
                    run([sys.executable, "scripts/wstats.py", "--mode", mode, "4096", "1024", "64"], e, log, 200)
                    print(open(log).read(), flush=True)


if __name__ == "__main__":
    main()
