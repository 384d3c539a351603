"""include/mp2g.h is usable from plain C: build examples/c_abi_demo.c with gcc, run it on the GPU and
compare the proof it serializes with the one the Python harness gets for the same inputs."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo():
    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + os.path.join(ROOT, "mapreduce-plonky2_amd"),
                           "-lmp2gpu", "-Wl,-rpath," + os.path.join(ROOT, "mapreduce-plonky2_amd"), "-o", exe])
    return exe


def test_header_compiles_as_c():
    """No GPU needed: the header is valid C11 and the demo links against the library."""
    build_demo()


@pytest.mark.gpu
def test_c_client_matches_python(ctx, mp2):
    exe = build_demo()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"proof_words=(\d+) bytes=(\d+) fnv1a=([0-9a-f]+) pow_witness=(\d+)", out.stdout)
    assert m, out.stdout
    ws = (5, 9, 4, 3)
    fp = mp2.standard_recursion_params(6, ws, pow_bits=6, num_queries=4)
    vals = [O.rand_field((w, 64), 100 + i) for i, w in enumerate(ws)]
    caps, openings, proof = mp2.pcs_prove(ctx, fp, vals, O.rand_field(4, 1), O.rand_field(4, 2))
    data = mp2.serialize_proof(fp, 2, caps, openings, proof, [7, 8, 9])
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(m.group(1)) == fp.proof_words and int(m.group(2)) == len(data)
    assert int(m.group(4)) == int(proof[-1])
    assert m.group(3) == f"{h:016x}"
