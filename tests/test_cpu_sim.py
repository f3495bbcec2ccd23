"""The per-occurrence device functions (cgx_amd/csrc/cgx_rules.h), built for the host with
ASan/UBSan, must reproduce the oracle's frequent-pair lists, rule arrays and MaxLex floats."""
import os
import subprocess

import oracle_py as op

ROOT = op.ROOT


def test_device_rule_functions_on_host(oracle_bin, tmp_path):
    exe = str(tmp_path / "sim_rules")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=alignment", "-std=c++17",
                    os.path.join(ROOT, "tests", "cpu_sim", "sim_rules.cpp"), "-o", exe], check=True)
    dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, os.path.join(ROOT, "tests", "golden", "tiny"), str(tmp_path / "o"), dump)
    r = subprocess.run([exe, dump], capture_output=True, text=True)
    assert r.returncode == 0 and "SIM OK" in r.stdout, r.stdout + r.stderr


def test_writer_float_formatting_equals_printf(tmp_path):
    exe = str(tmp_path / "f6test")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-w", os.path.join(ROOT, "tests", "cpu_sim", "f6test.c"), "-lm", "-lpthread",
                    "-Wl,--unresolved-symbols=ignore-all", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "F6 OK" in r.stdout, r.stdout


def test_formatter_gzip_members_inflate_to_the_plain_text(tmp_path):
    """cgx_amd/csrc/cgx_fmt.h built for the host with ASan/UBSan: the DEFLATE symbols the formatter emits for random lexicons
    (counting pass == writing pass, every gzip member inflated by zlib == the plain text of its emission group, the CRC fold ==
    zlib's crc32, the whole buffer == one multi-member stream), with ASCII and with non-ASCII spellings, short and long groups."""
    exe = str(tmp_path / "gz_sim")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=alignment", "-std=c++17",
                    os.path.join(ROOT, "tests", "cpu_sim", "gz_sim.cpp"), "-lz", "-o", exe], check=True)
    r = subprocess.run([exe, "8"], capture_output=True, text=True)
    assert r.returncode == 0 and "GZ SIM OK" in r.stdout, r.stdout + r.stderr
    # the bytes themselves (FNV-1a over every plain line / every deflate byte of the seeded cases): the values of the formatter that made
    # the golden grammar files on the GPU (cgx_fmt.h of commit 0e054e3 built against this simulator), recorded in round 4; word ids of one to ten digits among the cases
    assert "GZ SIM DIGEST plain b93b289b1bf666bc deflate 9b2105cfa957170e" in r.stdout, r.stdout


def test_window_transpose_algebra():
    """load_window / cgx_win_load (cgx_search.inc, cgx_extract.inc): the eight lanes of a group fetch the group's eight windows
    one after the other -- load j gives lane k piece k of window j -- and three butterfly stages (masks 4, 2, 1: a lane whose
    bit is set sends r[j] and receives into r[j], the others send r[j | m] and receive into r[j | m]) must leave lane k with
    piece p of ITS window in r[p].  The same stage rule, restated on an 8 x 8 array of labels; and the piece a backward walk can
    reach: memory piece k is piece 7 - k counted from the driving phrase."""
    r = [[(j, k) for k in range(8)] for j in range(8)]       # r[j][k] = (window, piece) held by lane k in register j
    for m in (4, 2, 1):
        new = [row[:] for row in r]
        for j in range(8):
            if j & m:
                continue
            for k in range(8):
                up = bool(k & m)
                send = lambda lane: r[j][lane] if (lane & m) else r[j | m][lane]        # what `lane` puts on the exchange
                got = send(k ^ m)
                if up:
                    new[j][k] = got
                else:
                    new[j | m][k] = got
        r = new
    for k in range(8):
        for p in range(8):
            assert r[p][k] == (k, p), (k, p, r[p][k])
    # backward window: memory entries base .. base + 15 with base = edge - 15; entry e of the walk is memory entry 15 - e, so
    # memory piece k (entries 2k, 2k + 1) holds walk entries 15 - 2k, 14 - 2k = piece 7 - k of the walk
    for k in range(8):
        assert {(15 - 2 * k) // 2, (14 - 2 * k) // 2} == {7 - k}
