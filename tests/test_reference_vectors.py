"""Reconciliation against the REFERENCE's own dependencies (plonky2 0.2.2 @ Lagrange-Labs/plonky2#upstream, poseidon2_plonky2,
plonky2_ecgfp5, mp2-common, mp2-v1): consumes `tests/golden/reference_vectors.json` (and `..._poseidon.json`, the reference's
`original_poseidon` build) written by the Rust dumper tools/ref_vectors/ (run INSIDE the reference's workspace -- this image has
no Rust toolchain, so the files do not exist yet and the reference-backed cases SKIP with that reason). Every "parity unpinned"
row of DESIGN.md section 2 has a check here: field constants, both permutations and their sponges, Merkle helpers, the column-id
formula, FFT ordering, PolynomialBatch leaves / cap / Merkle proof, the challenger, Ecgfp5 map-to-curve / add / scalar mul /
Weierstrass form, the off-chain half of table creation (row digest, off-chain data commitment and its update, cells-tree and
row-tree node hashes), one complete proof (bincode bytes, bit-exact given the reference's proof-of-work witness) and the first
wrapping step of the recursion framework over it (a 2^12-row recursive verifier, its full wire matrix and proof).

The consumer itself is exercised on every run with a file of the same schema made from THIS repository's oracle
(tools/ref_vectors/self_vectors.py): that proves the test reads the schema and drives every section -- not parity.
CPU: the oracle. `-m gpu`: the HIP library through its C ABI."""
import ctypes
import importlib
import json
import os
import re
import sys

import numpy as np
import pytest

import circuits as OC
import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FILES = {"reference (default config)": os.path.join(GOLDEN, "reference_vectors.json"),
             "reference (original_poseidon)": os.path.join(GOLDEN, "reference_vectors_poseidon.json")}
NO_REF = ("no reference vectors: run tools/ref_vectors/run.sh in a checkout of the reference with its Rust toolchain "
          "(none in this image) to create tests/golden/reference_vectors.json")
VARIANT = {"poseidon2": 0, "poseidon": 1}
MULT_GEN, TWO_GEN = 14293326489335486720, 7277203076849721926
PC = importlib.import_module("mapreduce-plonky2_amd.circuits")


@pytest.fixture(scope="module")
def self_vectors(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tools", "ref_vectors"))
    import self_vectors as SV
    out = {}
    for hasher in ("poseidon2", "poseidon"):
        # through a file: the JSON round trip is part of what is exercised (u64 values above 2^53 must survive it)
        path = tmp_path_factory.mktemp("vec") / f"self_{hasher}.json"
        path.write_text(json.dumps(SV.make(hasher)))
        out[hasher] = json.loads(path.read_text())
    return out


def _load(which, self_vectors):
    if which.startswith("self"):
        return self_vectors["poseidon" if "poseidon)" in which and "poseidon2" not in which else "poseidon2"]
    path = REF_FILES[which]
    if not os.path.exists(path):
        pytest.skip(NO_REF)
    with open(path) as f:
        return json.load(f)


SOURCES = ["self (poseidon2)", "self (poseidon)"] + list(REF_FILES)
u64 = lambda a: np.asarray(a, dtype=np.uint64)


# ---- gate ids -----------------------------------------------------------------------------------------------------------------
def parse_gate_id(s):
    """(kind, p0, p1, p2) of a plonky2 Gate::id() string (format!("{:?}") of the gate struct, some with a <D=..> / <WIDTH=..> tail)"""
    num = lambda key: int(re.search(key + r": (\d+)", s).group(1))
    if s.startswith("NoopGate"):
        return (PC.NOOP, 0, 0, 0)
    if s.startswith("ConstantGate"):
        return (PC.CONSTANT, num("num_consts"), 0, 0)
    if s.startswith("PublicInputGate"):
        return (PC.PUBLIC_INPUT, 0, 0, 0)
    if s.startswith("ArithmeticGate"):
        return (PC.ARITHMETIC, num("num_ops"), 0, 0)
    if s.startswith("ArithmeticExtensionGate"):
        return (PC.ARITHMETIC_EXT, num("num_ops"), 0, 0)
    if s.startswith("MulExtensionGate"):
        return (PC.MUL_EXT, num("num_ops"), 0, 0)
    if s.startswith("Poseidon2Gate"):
        return (PC.POSEIDON2, 0, 0, 0)
    if s.startswith("PoseidonMdsGate"):
        return (PC.POSEIDON_MDS, 0, 0, 0)
    if s.startswith("PoseidonGate"):
        return (PC.POSEIDON, 0, 0, 0)
    if s.startswith("BaseSumGate"):
        return (PC.BASE_SUM, num("num_limbs"), int(re.search(r"Base: (\d+)", s).group(1)), 0)
    if s.startswith("RandomAccessGate"):
        return (PC.RANDOM_ACCESS, num("bits"), num("num_copies"), num("num_extra_constants"))
    if s.startswith("ExponentiationGate"):
        return (PC.EXPONENTIATION, num("num_power_bits"), 0, 0)
    if s.startswith("ReducingExtensionGate"):
        return (PC.REDUCING_EXT, num("num_coeffs"), 0, 0)
    if s.startswith("ReducingGate"):
        return (PC.REDUCING, num("num_coeffs"), 0, 0)
    if s.startswith("CosetInterpolationGate"):
        return (PC.COSET_INTERPOLATION, num("subgroup_bits"), num("degree"), 0)
    raise ValueError(f"gate id not known to this test: {s}")


def circuit_of(p, variant):
    """a circuits.Circuit (what the oracle and the HIP prover take) from the `proof` section"""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    ckt = PC.Circuit()
    ckt.log_n = int(p["degree_bits"])
    gates = []
    for gid, si in zip(p["gates"], p["selector_indices"]):
        kind, p0, p1, p2 = parse_gate_id(gid)
        lo, hi = p["selector_groups"][si]
        gates.append(mp2.Gate(kind, p0, p1, p2, int(si), int(lo), int(hi)))
    ckt.gates, ckt.num_selectors = gates, len(p["selector_groups"])
    ckt.pre = u64(p["constants_sigmas"])
    ckt.wires = u64(p["wires"])
    ckt.num_constants = int(ckt.pre.shape[0]) - PC.NUM_ROUTED
    assert ckt.num_constants == ckt.num_selectors + int(p["num_constants"]) or ckt.num_constants == int(p["num_constants"]), "constants = selectors + gate constants"
    ckt.public_inputs = u64(p["public_inputs"])
    ckt.pi_hash = O.hash_n_to_m_no_pad(ckt.public_inputs, 4, variant)
    ckt.pi_row = None
    ckt.gate_array = (mp2.Gate * len(gates))(*gates)
    ckt.luts, ckt.num_lookup_selectors, ckt.num_lookup_polys = [], 0, 0
    ckt.domain_separator = []
    return ckt


# ---- CPU: the oracle against the vectors ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", SOURCES)
def test_field_constants(which, self_vectors):
    f = _load(which, self_vectors)["field"]
    assert int(f["order"]) == O.P and int(f["multiplicative_group_generator"]) == MULT_GEN == int(f["coset_shift"])
    assert int(f["power_of_two_generator"]) == TWO_GEN and int(f["two_adicity"]) == 32
    assert int(f["root_of_unity_log3"]) == pow(TWO_GEN, 1 << 29, O.P) == 1 << 24 and int(f["root_of_unity_log6"]) == pow(TWO_GEN, 1 << 26, O.P) == 8


@pytest.mark.parametrize("which", SOURCES)
def test_permutations_sponges_and_merkle_helpers(which, self_vectors):
    """Poseidon2 (the reference's default hasher -- unpinned until this file exists) and Poseidon: permute([0..12)), hash_no_pad at
    lengths 0 / 1 / 4 / 7 / 8 / 9 / 17 / 135, hash_pad, hash_or_noop around its 4-limb threshold, two_to_one"""
    v_all = _load(which, self_vectors)["hashers"]
    lib = O.lib()
    for name, v in VARIANT.items():
        h = v_all[name]
        assert O.perm(np.arange(12, dtype=np.uint64), v).tolist() == h["permute_0_to_11"], f"{name} permutation"
        for n, want in h["hash_no_pad"].items():
            assert O.hash_n_to_m_no_pad(np.arange(int(n), dtype=np.uint64), 4, v).tolist() == want, f"{name} hash_no_pad({n})"
        for key, fn, first in (("hash_pad", lib.orc_hash_pad, 0), ("hash_or_noop", lib.orc_hash_or_noop, 1)):
            for n, want in h[key].items():
                a, o = O.arr(np.arange(first, first + int(n), dtype=np.uint64)), np.zeros(4, dtype=np.uint64)
                fn(v, O.p(a), O.sz(a.size), O.p(o))
                assert o.tolist() == want, f"{name} {key}({n})"
        l, r, o = O.arr([1, 2, 3, 4]), O.arr([5, 6, 7, 8]), np.zeros(4, dtype=np.uint64)
        lib.orc_two_to_one(v, O.p(l), O.p(r), O.p(o))
        assert o.tolist() == h["two_to_one_1234_5678"]


@pytest.mark.parametrize("which", SOURCES)
def test_identifier_block_column(which, self_vectors):
    """mp2-v1/src/values_extraction/mod.rs:157-160: H(b"BLOCK_NUMBER" one byte per limb)[0] under the compiled configuration"""
    d = _load(which, self_vectors)
    limbs = np.frombuffer(b"BLOCK_NUMBER", dtype=np.uint8).astype(np.uint64)
    assert int(O.hash_n_to_m_no_pad(limbs, 4, VARIANT[d["default_hasher"]])[0]) == int(d["identifier_block_column"])


@pytest.mark.parametrize("which", SOURCES)
def test_fft_ordering(which, self_vectors):
    """plonky2_field fft / ifft / coset_fft / lde + coset_fft at 2^3 and 2^10 on the SplitMix64 stream (seed 0xC0FFEE02)"""
    for log_n, s in _load(which, self_vectors)["fft"].items():
        n = 1 << int(log_n)
        x = O.rand_field((1, n), 0xC0FFEE02)
        assert x[0].tolist() == s["input"], "the input stream (splitmix_field of the dumper = rand_field here)"
        assert O.fft(x)[0].tolist() == s["fft"] and O.fft(x, inverse=True)[0].tolist() == s["ifft"]
        assert O.fft(x, coset_shift=MULT_GEN)[0].tolist() == s["coset_fft"]
        padded = np.concatenate([x, np.zeros((1, n), dtype=np.uint64)], axis=1)
        assert O.fft(padded, coset_shift=MULT_GEN)[0].tolist() == s["lde1_coset_fft"]


@pytest.mark.parametrize("which", SOURCES)
def test_polynomial_batch_layout(which, self_vectors):
    """PolynomialBatch::from_values: coefficients, leaf i = evaluations at g w^bitrev(i), the Merkle cap, one Merkle proof"""
    d = _load(which, self_vectors)
    b, v = d["polynomial_batch"], VARIANT[d["default_hasher"]]
    values = u64(b["values"])
    assert values.ravel().tolist() == O.rand_field(values.size, 0xC0FFEE02).tolist()
    coeffs = O.fft(values, inverse=True)
    assert coeffs.tolist() == b["coeffs"]
    leaves = O.lde_leaves(coeffs, b["rate_bits"])
    for i, leaf in zip(b["leaf_indices"], b["leaves"]):
        assert leaves[i].tolist() == leaf, f"leaf {i}"
    levels = O.merkle_build(leaves, b["cap_height"], v)
    assert O.merkle_cap(levels, b["cap_height"]).reshape(-1, 4).tolist() == b["cap"]
    sib = O.merkle_prove(levels, b["log_n"] + b["rate_bits"], b["cap_height"], b["proof_index"])
    assert sib.reshape(-1, 4).tolist() == b["proof_siblings"]


@pytest.mark.parametrize("which", SOURCES)
def test_challenger_script(which, self_vectors):
    sys.path.insert(0, os.path.join(ROOT, "tools", "ref_vectors"))
    import self_vectors as SV
    d = _load(which, self_vectors)
    got = SV.challenger_script(VARIANT[d["default_hasher"]])
    for key in ("first_two", "extension", "next_nine"):
        assert got[key] == d["challenger"][key], key


@pytest.mark.parametrize("which", SOURCES)
def test_ecgfp5_group(which, self_vectors):
    """map_to_curve_point, point addition / doubling, scalar multiplication and the 11-limb Weierstrass form (the y sign convention
    of to_weierstrass is one of the unpinned rows) against plonky2_ecgfp5 through mp2-common's wrappers"""
    sys.path.insert(0, os.path.join(ROOT, "tools", "ref_vectors"))
    import self_vectors as SV
    d = _load(which, self_vectors)
    want, got = d["ecgfp5"], SV.ecgfp5_section(VARIANT[d["default_hasher"]])
    for a, b in zip(got["map_to_curve"], want["map_to_curve"]):
        assert a["input"] == b["input"] and a["point"] == b["point"], "map_to_curve_point"
    for key in ("add_0_1", "sum_all", "double_0", "neutral", "scalar_mul_0"):
        assert got[key] == want[key], key
    assert got["hash_to_int"]["value"] == str(want["hash_to_int"]["value"]) and got["hash_to_int"]["flatten"] == want["hash_to_int"]["flatten"]


@pytest.mark.parametrize("which", SOURCES)
def test_table_digest_commitment_and_cell_hashes(which, self_vectors):
    """mp2-v1's off-chain half of table creation on the file's own six rows: row_unique_data, compute_table_row_digest
    (values_extraction/mod.rs:499-571), off_chain_data_commitment and its incremental update (api.rs:556-612: groups of equal primary
    values in increasing order, add_primary_index_to_digest, the flattened hash chain, 32 little-endian u32 bytes) and
    MerkleCell::aggregate (indexing/cell.rs:120-157) and RowPayload::aggregate (indexing/row.rs:257-317: the hash and the min / max of
    the secondary index) with no, one and two children"""
    sys.path.insert(0, os.path.join(ROOT, "tools", "ref_vectors"))
    import self_vectors as SV
    d = _load(which, self_vectors)
    if "table" not in d:
        pytest.skip("a vector file written before the `table` section was added to the dumper")
    want = d["table"]
    got = SV.table_section(VARIANT[d["default_hasher"]], given=want)
    for key in ("row_unique_data_row0", "row_digest", "commitment_rows_0_to_3", "commitment_updated_with_rows_4_5", "cells_tree", "row_tree"):
        assert got[key] == want[key], key
    # the dumper's rows come from the SplitMix64 stream this repository's workload generator uses: the same rows are made here
    assert SV.table_section(VARIANT[d["default_hasher"]])["rows"] == want["rows"], "rand_field(96, 0xC0FFEE04), four u64 words per value, least significant first"


@pytest.mark.parametrize("section", ["proof", "proof_recursive"])
@pytest.mark.parametrize("which", SOURCES)
def test_one_complete_proof(which, section, self_vectors):
    """`proof`: the reference's proof of a 2^5-row circuit; `proof_recursive`: of the first wrapping step over it (wrap_circuit.rs:64-99,
    a 2^12-row recursive verifier: Poseidon2 / BaseSum / RandomAccess / Reducing / ArithmeticExtension ... rows). (i) the wire format -- csrc/wire.hip parses the reference's bincode bytes and writes
    them back identically; (ii) verifier data -- constants_sigmas cap and circuit digest from the preprocessed polynomials; (iii) the
    oracle's verifier accepts the reference's proof; (iv) the oracle's prove() of the same witness, given the reference's proof-of-work
    witness, IS the reference's proof: caps, openings, every FRI word"""
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    FW = importlib.import_module("mapreduce-plonky2_amd.framework")
    d = _load(which, self_vectors)
    if section not in d:
        pytest.skip(f"no `{section}` in this file (self-made vectors carry the recursive one for the Poseidon2 configuration only: the in-circuit verifier of this repository hashes with Poseidon2)")
    p, v = d[section], VARIANT[d["default_hasher"]]
    ckt = circuit_of(p, v)
    assert [pow(MULT_GEN, j, O.P) for j in range(PC.NUM_ROUTED)] == [int(x) for x in p["k_is"]]
    fp_mp2 = FW.circuit_fri_params(ckt, v)
    raw = bytes.fromhex(p["proof_bincode_hex"])
    caps, openings, fri, pis = mp2.deserialize_proof(fp_mp2, ckt.num_constants, raw, len(p["public_inputs"]))
    assert pis.tolist() == [int(x) for x in p["public_inputs"]] and int(fri[-1]) == int(p["pow_witness"])
    assert mp2.serialize_proof(fp_mp2, ckt.num_constants, caps, openings, fri, pis) == raw, "bincode bytes: parse + write = identity"
    cap = O.merkle_cap(O.merkle_build(O.lde_leaves(O.fft(ckt.pre, inverse=True), 3), 4, v), 4)
    assert cap.reshape(-1, 4).tolist() == p["constants_sigmas_cap"]
    dom, e = np.zeros(4, dtype=np.uint64), O.arr(np.zeros(0, dtype=np.uint64))
    O.lib().orc_hash_pad(v, O.p(e), O.sz(0), O.p(dom))
    digest = O.hash_n_to_m_no_pad(list(cap.reshape(-1)) + list(dom) + [ckt.log_n], 4, v)
    assert digest.tolist() == [int(x) for x in p["circuit_digest"]], "circuit digest = H(cap || H_pad([]) || degree_bits)"
    fp = OC.oracle_params(ckt, variant=v)
    caps[0] = cap.reshape(-1)  # the constants_sigmas cap is verifier data, not part of the proof's bytes
    assert OC.verify(ckt, fp, digest, ckt.pi_hash, caps, openings, fri) == 0, "the oracle's verifier on the reference's proof"
    O.lib().orc_set_pow_witness(ctypes.c_uint64(int(p["pow_witness"])))
    ocaps, oopen, oproof, _ = OC.prove_witness(ckt, fp, digest, ckt.wires, ckt.pi_hash)
    assert np.array_equal(ocaps[1:], caps[1:]), "wires / Z / quotient caps"
    assert np.array_equal(oopen, openings), "openings at zeta and g zeta"
    assert np.array_equal(oproof, fri), "FRI proof (commit-phase caps, query rounds, final polynomial, PoW witness)"


def test_gate_id_strings_parse():
    ids = {"NoopGate": (PC.NOOP, 0, 0, 0), "ConstantGate { num_consts: 2 }": (PC.CONSTANT, 2, 0, 0), "PublicInputGate": (PC.PUBLIC_INPUT, 0, 0, 0),
           "ArithmeticGate { num_ops: 20 }": (PC.ARITHMETIC, 20, 0, 0), "BaseSumGate { num_limbs: 63 } + Base: 2": (PC.BASE_SUM, 63, 2, 0),
           "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>": (PC.RANDOM_ACCESS, 4, 4, 2),
           "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>": (PC.POSEIDON, 0, 0, 0),
           "Poseidon2Gate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>": (PC.POSEIDON2, 0, 0, 0),
           "ReducingExtensionGate { num_coeffs: 32 }": (PC.REDUCING_EXT, 32, 0, 0), "ReducingGate { num_coeffs: 43 }": (PC.REDUCING, 43, 0, 0)}
    for s, want in ids.items():
        assert parse_gate_id(s) == want


# ---- GPU: the HIP library against the same vectors ------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("which", SOURCES)
def test_hip_library_against_the_vectors(which, self_vectors, ctx, mp2):
    """the product path, through the C ABI: sponges, NTT orderings, PolynomialBatch commitment and Merkle proof, challenger,
    Ecgfp5, and prove() of the reference's witness (equal to the reference's proof up to the proof-of-work witness: the HIP prover
    returns the smallest; the oracle test above closes the gap given the witness)"""
    FW = importlib.import_module("mapreduce-plonky2_amd.framework")
    d = _load(which, self_vectors)
    dv = VARIANT[d["default_hasher"]]
    for name, v in VARIANT.items():
        h = d["hashers"][name]
        for n, want in h["hash_no_pad"].items():
            if int(n):
                assert ctx.hash_no_pad_batch(np.arange(int(n), dtype=np.uint64).reshape(1, -1), 4, v)[0].tolist() == want, f"{name} hash_no_pad({n})"
        assert [int(x) for x in ctx.hash_no_pad([1, 2, 3, 4, 5, 6, 7, 8], v)] == h["two_to_one_1234_5678"]
    for log_n, s in d["fft"].items():
        x = u64(s["input"]).reshape(1, -1)
        assert ctx.ntt(x)[0].tolist() == s["fft"] and ctx.ntt(x, inverse=True)[0].tolist() == s["ifft"]
        assert ctx.ntt(x, coset_shift=MULT_GEN)[0].tolist() == s["coset_fft"]
        padded = np.concatenate([x, np.zeros_like(x)], axis=1)
        assert ctx.ntt(padded, coset_shift=MULT_GEN)[0].tolist() == s["lde1_coset_fft"]
    b = d["polynomial_batch"]
    pb = mp2.PolynomialBatch.from_values(ctx, u64(b["values"]), b["rate_bits"], b["cap_height"], dv)
    assert pb.coeffs.tolist() == b["coeffs"] and pb.cap.reshape(-1, 4).tolist() == b["cap"]
    leaves, sibs = pb.open(b["leaf_indices"])
    assert leaves.tolist() == b["leaves"], "leaf i = evaluations at g w^bitrev(i)"
    assert sibs[b["leaf_indices"].index(b["proof_index"])].tolist() == b["proof_siblings"]
    pb.free()
    ch = mp2.Challenger(ctx, dv)
    ch.observe_elements([1, 2, 3])
    first = [int(x) for x in ch.get_n_challenges(2)[0]]
    ch.observe_elements([7, 8, 9, 10])
    ext = [int(x) for x in ch.get_n_challenges(2)[0]]
    ch.observe_elements(list(range(11, 23)))
    nine = [int(x) for x in ch.get_n_challenges(9)[0]]
    ch.free()
    assert (first, ext, nine) == (d["challenger"]["first_two"], d["challenger"]["extension"], d["challenger"]["next_nine"])
    e = d["ecgfp5"]
    ws = []
    for m in e["map_to_curve"]:
        w, wei = mp2.map_to_curve_batch(ctx, u64(m["input"]).reshape(1, -1), dv, weierstrass=True)
        assert w[0].tolist() == m["point"]["encode"] and wei[0].tolist() == m["point"]["fields"]
        ws.append(w[0])
    for key, pts in (("add_0_1", ws[:2]), ("sum_all", ws), ("double_0", [ws[0], ws[0]])):
        w, wei = mp2.curve_sum(ctx, np.stack(pts), weierstrass=True)
        assert w.tolist() == e[key]["encode"] and wei.tolist() == e[key]["fields"], key
    w, wei = mp2.scalar_mul_batch(ctx, ws[0].reshape(1, 5), [int(e["hash_to_int"]["value"])], weierstrass=True)
    assert w[0].tolist() == e["scalar_mul_0"]["encode"] and wei[0].tolist() == e["scalar_mul_0"]["fields"]
    # the off-chain half of table creation: row digest, commitment chain, cells-tree node hashes
    if "table" in d:
        sys.path.insert(0, os.path.join(ROOT, "tools", "ref_vectors"))
        import self_vectors as SV
        DG = importlib.import_module("mapreduce-plonky2_amd.digest")
        IX = importlib.import_module("mapreduce-plonky2_amd.indexing")
        t = d["table"]
        ids, uq = [int(x) for x in t["column_ids"]], [int(x) for x in t["row_unique_columns"]]
        values = np.array([[SV.u256_words(x) for x in r["values"]] for r in t["rows"]], dtype=np.uint32)
        primary = np.array([SV.u256_words(r["primary"]) for r in t["rows"]], dtype=np.uint32)
        unique = np.ascontiguousarray(values[:, [ids.index(u_) for u_ in uq]])
        w, wei = mp2.compute_table_row_digest(ctx, ids, values, unique, dv)
        assert w.tolist() == t["row_digest"]["encode"] and wei.tolist() == t["row_digest"]["fields"], "compute_table_row_digest"
        assert [int(x) for x in ctx.hash_no_pad(unique[0].reshape(-1).astype(np.uint64), dv)] == t["row_unique_data_row0"]
        first = DG.off_chain_data_commitment(ctx, int(t["primary_id"]), primary[:4], ids, values[:4], uq, dv)
        assert bytes(first).hex() == t["commitment_rows_0_to_3"], "off_chain_data_commitment"
        second = DG.update_off_chain_data_commitment(ctx, int(t["primary_id"]), primary[4:], ids, values[4:], uq, bytes(first), dv)
        assert bytes(second).hex() == t["commitment_updated_with_rows_4_5"], "update_off_chain_data_commitment"
        empty = IX.empty_poseidon_hash(ctx, dv)
        v0 = [int(x) for x in t["rows"][0]["values"]]
        hexof = lambda h: np.asarray(h, dtype="<u8").tobytes().hex()
        leaves = IX.cell_node_hashes(ctx, [empty, empty], [empty, empty], [ids[1], ids[3]], [v0[1], v0[3]], dv)
        assert [hexof(leaves[0]), hexof(leaves[1])] == [t["cells_tree"]["leaf_column_1"], t["cells_tree"]["leaf_column_3"]]
        above = IX.cell_node_hashes(ctx, [leaves[0], leaves[0]], [empty, leaves[1]], [ids[2], ids[2]], [v0[2], v0[2]], dv)
        assert [hexof(above[0]), hexof(above[1])] == [t["cells_tree"]["column_2_over_left_child"], t["cells_tree"]["column_2_over_both"]]
        # row-tree nodes: the secondary index is the first of the other columns; min / max are the file's (the CPU test derives them)
        rt, sec = t["row_tree"], [int(r["values"][0]) for r in t["rows"]]
        croot = [above[1]]
        la = IX.row_node_hashes(ctx, [empty], [empty], [sec[0]], [sec[0]], [ids[0]], [sec[0]], croot, dv)[0]
        lb = IX.row_node_hashes(ctx, [empty], [empty], [sec[2]], [sec[2]], [ids[0]], [sec[2]], croot, dv)[0]
        assert [hexof(la), hexof(lb)] == [rt["leaf_row_0"]["hash"], rt["leaf_row_2"]["hash"]]
        for key, (l_, r_) in (("row_1_over_left_child", (la, empty)), ("row_1_over_right_child", (empty, lb)), ("row_1_over_both", (la, lb))):
            h = IX.row_node_hashes(ctx, [l_], [r_], [int(rt[key]["min"])], [int(rt[key]["max"])], [ids[0]], [sec[1]], croot, dv)[0]
            assert hexof(h) == rt[key]["hash"], key
    # prove() of the reference's witnesses: the small circuit and the recursive verifier over its proof
    for section in ("proof", "proof_recursive"):
        if section in d:
            _hip_prove_section(ctx, mp2, FW, d[section], dv)


def _hip_prove_section(ctx, mp2, FW, p, dv):
    ckt = circuit_of(p, dv)
    cp = FW.CircuitProver(ctx, ckt, 1, dv, witness_check=True)
    assert cp.constants_sigmas_cap.reshape(-1, 4).tolist() == p["constants_sigmas_cap"] and [int(x) for x in cp.circuit_digest] == [int(x) for x in p["circuit_digest"]]
    cp.prove(ctx.to_device(ckt.wires[None]), ctx.to_device(ckt.pi_hash[None]))
    assert cp.pr.witness_status().tolist() == [0], "the reference's witness satisfies every gate and copy constraint on the device"
    caps, openings, proofs = cp.results()
    rcaps, ropen, rfri, _ = mp2.deserialize_proof(cp.fp, ckt.num_constants, bytes.fromhex(p["proof_bincode_hex"]), len(p["public_inputs"]))
    assert np.array_equal(caps[0][1:], rcaps[1:]) and np.array_equal(openings[0], ropen), "caps and openings do not depend on the PoW witness"
    fp = OC.oracle_params(ckt, variant=dv)
    n_before_queries = fp.n_layers * (4 << fp.cap_height)
    assert np.array_equal(proofs[0][:n_before_queries], rfri[:n_before_queries]), "FRI commit-phase caps"
    n_final = 2 << (fp.log_n - sum(fp.arity_bits[i] for i in range(fp.n_layers)))
    assert np.array_equal(proofs[0][-1 - n_final:-1], rfri[-1 - n_final:-1]), "FRI final polynomial"
    if int(proofs[0][-1]) == int(rfri[-1]):
        assert np.array_equal(proofs[0], rfri), "same PoW witness: the whole proof"
    cp.free()
