"""One-off randomized parity sweep on a GPU box: seeded fixtures of several shapes, CPU oracle vs HIP product,
sha256 over all grammar files.  usage: python tools/stress_parity.py [--cases N]  (needs oracle/strmatch_oracle built)."""
import argparse, hashlib, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_fixture


def sha_dir(d, nq):
    h = hashlib.sha256()
    for q in range(nq):
        with open(os.path.join(d, "grammar.%d.s" % q), "rb") as f:
            h.update(f.read()); h.update(b"\0")
    return h.hexdigest()


SHAPES = [  # pairs, vocab, queries, seed, lo, hi, options
    (20000, 500, 80, 101, 4, 30, {}),
    (3000, 104, 60, 102, 6, 40, {}),                          # tiny vocabulary: dense matches, long occurrence lists; its lex file has a NULL NULL row
    (8000, 120, 70, 103, 3, 25, {"pool_cap": 16, "look_rec_cap": 8}),
    (30000, 2000, 120, 104, 5, 35, {"sub_batch": 37, "async_write": 1}),
    (6000, 150, 50, 105, 10, 60, {"chunk_items": 4096, "use_lex_hash": 0}),
    (12000, 110, 90, 106, 4, 28, {"append_guess_milli": 1, "append_slack": 0, "wide_hits2": 1}),
    (15000, 300, 100, 107, 2, 12, {"device_format": 0}),      # short sentences, host formatter
    (2500, 101, 40, 108, 20, 80, {"use_bigrams": 0}),         # long sentences, hardly more than the 100 frequent tokens
    (5000, 200, 90, 109, 3, 20, {"auto_batch_tokens": 64, "async_write": 1}),   # automatic internal batches of a few queries each
    (9000, 130, 80, 110, 4, 30, {"ngram_tables": 3, "use_layouts": 0}),         # l = 4, 5 by nested search; plain ltar / rtar tables in the tightness test
    (25000, 101, 130, 41001, 15, 18, {"async_write": 1, "device_format": 0, "auto_batch_tokens": 40, "pool_cap": 64, "look_rec_cap": 5}),   # host formatter, many tiny batches in flight:
                                                                                # caught the page-locked lexicon arena of batch k being refilled by batch k+2 while k was still being written
]


def run_case(cgx_amd, shape, verbose=True):
    """Generates the fixture, runs oracle and product, returns (identical, summary line)."""
    pairs, vocab, nq, seed, lo, hi, opts = shape
    oracle = os.path.join(ROOT, "oracle", "strmatch_oracle")
    tmp = tempfile.mkdtemp(prefix="cgx_stress_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        fx = os.path.join(tmp, "fx"); gen_fixture.write_fixture(fx, pairs, vocab, nq, seed, lo=lo, hi=hi)
        files = [os.path.join(fx, n) for n in ("corpus.f", "query.f", "corpus.e", "corpus.a", "lex.txt")]
        od, pd = os.path.join(tmp, "o"), os.path.join(tmp, "p"); os.mkdir(od); os.mkdir(pd)
        long_mode = bool(opts.get("long_sentences"))                # not a library option: both sides run with their long-sentence switch
        t0 = time.time(); subprocess.run([oracle] + (["--long-sentences"] if long_mode else []) + files + [od], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); t_or = time.time() - t0
        ex = cgx_amd.Extractor(0)
        for k, v in opts.items():
            if k != "long_sentences": ex.set_option(k, int(v))
        corpus = cgx_amd.Corpus.load(files[0], files[2], files[3], files[4], long_sentences=long_mode); ex.upload_corpus(corpus)
        t0 = time.time(); n = ex.extract_grammars(corpus, files[1], pd); ex.flush(); t_gpu = time.time() - t0
        c = ex.counts()
        same = sha_dir(od, nq) == sha_dir(pd, nq)
        line = "pairs %6d vocab %5d queries %4d opts %-60s rules %9d h1 %9d h2 %9d oracle %.1fs gpu %.2fs %s" % (pairs, vocab, nq, opts, n, c["h1"], c["h2"], t_or, t_gpu, "OK" if same else "MISMATCH")
        if not same and verbose:                                # show where the first difference is
            for q in range(nq):
                a = open(os.path.join(od, "grammar.%d.s" % q), "rb").read().split(b"\n"); b = open(os.path.join(pd, "grammar.%d.s" % q), "rb").read().split(b"\n")
                if a != b:
                    k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                    line += "\n  first difference: query %d line %d of %d (oracle) / %d (product)" % (q, k, len(a), len(b))
                    line += "\n  oracle : %r\n  product: %r" % (a[k] if k < len(a) else None, b[k] if k < len(b) else None)
                    sa_, sb_ = set(a), set(b)
                    line += "\n  lines only in oracle: %d, only in product: %d" % (len(sa_ - sb_), len(sb_ - sa_))
                    line += "\n  query: " + open(files[1]).read().split("\n")[q]
                    break
        ex.close(); corpus.close()
        return same, line
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--fuzz", type=int, default=0, help="instead of the fixed shapes: this many random shapes and option sets")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="with --fuzz: corpora of 50-150 k sentence pairs, 100-300 queries (tens of seconds of oracle time each)")
    ap.add_argument("--opt", action="append", default=[], help="name=value, overrides the case's options")
    args = ap.parse_args()
    import torch; torch.zeros(1, device="cuda:0")
    import cgx_amd; cgx_amd.load_library()
    shapes = SHAPES if args.only < 0 else [SHAPES[args.only]]
    if args.fuzz:
        import random
        r = random.Random(args.seed); shapes = []
        menu = {"pool_cap": [1, 7, 64], "look_rec_cap": [0, 1, 5, 40], "sub_batch": [1, 13, 50], "async_write": [1], "chunk_items": [1024, 4096, 1 << 16],
                "use_lex_hash": [0], "lex_flat": [0], "lex_bits": [0], "win_table": [1], "append_guess_milli": [1, 300], "append_slack": [0, 5], "wide_hits2": [1], "hit_order": [1], "device_format": [0], "use_bigrams": [0], "ngram_tables": [1, 2, 3, 4], "use_layouts": [0], "src_blocks": [0], "k1_limit": [128], "auto_batch_tokens": [40, 300], "lex_hash_bits": [3, 12, 20]}
        for i in range(args.fuzz):
            lo = r.choice([1, 2, 4, 8, 15]); hi = lo + r.choice([3, 10, 25, 60])
            opts = {k: r.choice(v) for k, v in menu.items() if r.random() < 0.25}
            if not args.big and r.random() < 0.2:               # long-sentence mode, half of the time on sentences the reference would refuse
                opts["long_sentences"] = 1
                if r.random() < 0.5:
                    shapes.append((r.choice([300, 800]), r.choice([130, 900]), r.choice([10, 25]), 1000 * args.seed + i, 200, r.choice([260, 330]), opts)); continue
            if args.big: shapes.append((r.choice([50000, 150000]), r.choice([150, 1000, 30000]), r.choice([100, 300]), 1000 * args.seed + i, lo, hi, opts))
            else: shapes.append((r.choice([1500, 4000, 9000, 25000]), r.choice([101, 105, 130, 250, 900, 5000]), r.choice([20, 60, 130]), 1000 * args.seed + i, lo, hi, opts))
    if args.opt: shapes = [s_[:6] + (dict(o.split("=") for o in args.opt),) for s_ in shapes]
    bad = 0
    for shape in shapes:
        same, line = run_case(cgx_amd, shape)
        print(line, flush=True); bad += not same
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
