#!/usr/bin/env python3
"""Mutation sweep over the text loaders and the corpus cache (host C, no GPU).

The four corpus files and the cache file are input a user hands to the product: whatever is in them, a loader has to come
back with a corpus or with an error message -- never with a fault.  Run against a sanitizer build of the host code:

    tools/asan_host.sh python3 tools/fuzz_loaders.py --cases 400 --seed 7

(`asan_host.sh` builds cgx_host.c with -fsanitize=address,undefined, links it to the device object as the product is linked,
preloads the sanitizer runtimes and sets CGX_LIB.)  Exit status 0: every case returned; the sanitizers abort the process on a finding.
"""
import argparse
import os
import random
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mutate(data: bytes, rng: random.Random) -> bytes:
    b = bytearray(data)
    for _ in range(rng.choice((1, 1, 2, 4))):
        kind = rng.randrange(12)
        pos = rng.randrange(len(b) + 1) if b else 0
        if kind == 0 and b:                                   # flip a byte
            b[pos % len(b)] = rng.randrange(256)
        elif kind == 1 and b:                                 # cut the tail (no final newline, half a token)
            del b[pos:]
        elif kind == 2:                                       # insert white space / newlines / CR
            b[pos:pos] = rng.choice((b" ", b"\n", b"\n\n", b"\r\n", b"\t", b"  \n", b"\0"))
        elif kind == 3 and b:                                 # drop a line
            s = bytes(b).split(b"\n"); del s[rng.randrange(len(s))]; b = bytearray(b"\n".join(s))
        elif kind == 4 and b:                                 # repeat a line many times (long file) or make one line very long
            s = bytes(b).split(b"\n"); i = rng.randrange(len(s))
            if rng.random() < 0.5:
                s[i:i] = [s[i]] * rng.choice((3, 50))
            else:
                s[i] = (s[i] + b" ") * rng.choice((20, 300))
            b = bytearray(b"\n".join(s))
        elif kind == 5:                                       # numbers the parsers do not expect
            b[pos:pos] = rng.choice((b" 999999999999-3 ", b" -1-2 ", b" 3- ", b" -", b" 1e309 ", b" nan inf ", b" 4294967296-0 ", b" 255-255 ", b" 0-255 ", b" 300-1 "))
        elif kind == 6:
            b = bytearray()                                   # empty file
        elif kind == 7 and b:                                 # a run of one byte
            b[pos:pos] = bytes([rng.randrange(256)]) * rng.choice((1, 64, 5000))
        elif kind == 8 and b:                                 # swap two lines
            s = bytes(b).split(b"\n"); i, j = rng.randrange(len(s)), rng.randrange(len(s)); s[i], s[j] = s[j], s[i]; b = bytearray(b"\n".join(s))
        elif kind == 9 and b:                                 # non-ASCII bytes inside a token
            b[pos:pos] = rng.choice((b"\xc3\xa9", b"\xff\xfe", b"\xe2\x80\x8b"))
        elif kind == 10 and b:                                # truncate to a prefix of whole lines
            s = bytes(b).split(b"\n"); b = bytearray(b"\n".join(s[:rng.randrange(len(s) + 1)]))
        elif kind == 11:
            b[pos:pos] = b"NULL NULL 0.5\n" if rng.random() < 0.5 else b"a b\n"
    return bytes(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fixture", default=os.path.join(ROOT, "tests", "golden", "tiny"))
    args = ap.parse_args()
    import cgx_amd
    cgx_amd.load_library()
    rng = random.Random(args.seed)
    names = ("corpus.f", "corpus.e", "corpus.a", "lex.txt")
    orig = {n: open(os.path.join(args.fixture, n), "rb").read() for n in names}
    work = tempfile.mkdtemp(prefix="cgx_fuzz_")
    loaded = failed = cache_ok = cache_bad = 0
    try:
        paths = {n: os.path.join(work, n) for n in names}
        # the cache of the untouched corpus, mutated below
        for n in names:
            open(paths[n], "wb").write(orig[n])
        c = cgx_amd.Corpus.load(paths["corpus.f"], paths["corpus.e"], paths["corpus.a"], paths["lex.txt"])
        good_cache = os.path.join(work, "good.cgx"); c.save(good_cache); want = c.checksum(); c.close()
        cache_bytes = open(good_cache, "rb").read()
        for case in range(args.cases):
            os.environ["CGX_THREADS"] = str(rng.choice((1, 2, 3, 8)))
            os.environ["CGX_LOAD_PIECE_MIN"] = str(rng.choice((1, 16, 200, 1 << 30)))
            victims = rng.sample(names, rng.choice((1, 1, 1, 2)))
            for n in names:
                open(paths[n], "wb").write(mutate(orig[n], rng) if n in victims else orig[n])
            long_mode = rng.random() < 0.25
            try:
                c = cgx_amd.Corpus.load(paths["corpus.f"], paths["corpus.e"], paths["corpus.a"], paths["lex.txt"], long_sentences=long_mode)
                c.checksum()
                if not long_mode and rng.random() < 0.3:          # what loaded must survive a save / load round trip
                    p = os.path.join(work, "rt.cgx"); s0 = c.checksum(); c.save(p)
                    c2 = cgx_amd.Corpus.load_cache(p); assert c2.checksum() == s0, "cache round trip changed the corpus (case %d)" % case; c2.close()
                c.close(); loaded += 1
            except cgx_amd.CgxError:
                failed += 1
            # the cache file: flipped bytes, truncations, grown tails
            m = bytearray(cache_bytes); k = rng.randrange(4)
            if k == 0:
                for _ in range(rng.choice((1, 3, 20))):
                    m[rng.randrange(len(m))] = rng.randrange(256)
            elif k == 1:
                del m[rng.randrange(len(m)):]
            elif k == 2:
                m += bytes(rng.randrange(256) for _ in range(rng.choice((1, 100))))
            else:                                             # header fields: plausible and absurd sizes
                off = rng.randrange(0, min(len(m) - 8, 256), 4); m[off:off + 4] = rng.choice((b"\xff\xff\xff\xff", b"\x00\x00\x00\x80", b"\x01\x00\x00\x00", b"\x00\x00\x00\x00"))
            p = os.path.join(work, "bad.cgx"); open(p, "wb").write(bytes(m))
            try:
                c = cgx_amd.Corpus.load_cache(p)
                assert c.checksum() == want, "a damaged cache was accepted with other contents (case %d)" % case     # e.g. a flip in padding, or a field rewritten with its own value
                c.close(); cache_ok += 1
            except cgx_amd.CgxError:
                cache_bad += 1
        print("fuzz_loaders: %d cases; text files: %d loaded, %d refused; cache files: %d accepted, %d refused" % (args.cases, loaded, failed, cache_ok, cache_bad))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
