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
