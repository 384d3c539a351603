"""Host logic that needs no GPU: bincode proof wire format (round trip, layout), sharding plans,
FRI parameter derivation."""
import ctypes
import struct

import numpy as np
import pytest

import oracle as O


def make_fp(mp2, log_n=6, ws=(5, 9, 4, 3)):
    ofp = O.standard_params(log_n, ws, pow_bits=4, num_queries=3)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    return ofp, fp


def test_params_match_oracle(mp2):
    for k in (5, 6, 12, 13, 14, 15):
        ofp = O.standard_params(k)
        fp = mp2.standard_recursion_params(k)
        assert bytes(fp) == bytes(ofp)
        assert fp.proof_words == O.lib().orc_fri_proof_words(ctypes.byref(ofp))
        assert fp.n_openings == O.lib().orc_n_openings(ctypes.byref(ofp))


def test_proof_wire_round_trip_and_layout(mp2):
    ofp, fp = make_fp(mp2)
    n = 1 << fp.log_n
    vals = [O.rand_field((w, n), 40 + i) for i, w in enumerate((5, 9, 4, 3))]
    cd, ph = O.rand_field(4, 1), O.rand_field(4, 2)
    caps, openings, proof = O.pcs_prove(ofp, vals, cd, ph)  # the oracle as the producer of a valid proof
    pis = O.rand_field(7, 3)
    data = mp2.serialize_proof(fp, 2, caps, openings, proof, pis)
    # bincode: first field is wires_cap = Vec<HashOut>: u64 length 16 then 64 limbs
    assert struct.unpack_from("<Q", data, 0)[0] == 16
    assert np.array_equal(np.frombuffer(data, dtype="<u8", count=64, offset=8), caps[1])
    # trailing: public_inputs Vec<F>
    assert np.array_equal(np.frombuffer(data[-56:], dtype="<u8"), pis)
    assert struct.unpack_from("<Q", data, len(data) - 64)[0] == 7
    c2, o2, p2, pi2 = mp2.deserialize_proof(fp, 2, data, 7)
    assert np.array_equal(c2[1:], caps[1:]) and np.array_equal(o2, openings) and np.array_equal(p2, proof)
    assert np.array_equal(pi2, pis)
    # a deserialized proof still verifies
    c2[0] = caps[0]
    assert O.pcs_verify(ofp, cd, ph, c2, o2, p2) == 0
    # malformed inputs are rejected
    with pytest.raises(mp2.Mp2gError):
        mp2.deserialize_proof(fp, 2, data[:-8], 7)
    bad = bytearray(data)
    bad[8:16] = struct.pack("<Q", 0xFFFFFFFFFFFFFFFF)  # non-canonical field element
    with pytest.raises(mp2.Mp2gError):
        mp2.deserialize_proof(fp, 2, bytes(bad), 7)


def test_proof_with_vk_blob(mp2):
    proof = bytes(range(40))
    cap = O.rand_field((16, 4), 1)
    dig = O.rand_field(4, 2)
    out = mp2.serialize_proof_with_vk(proof, cap, dig)
    assert out[:40] == proof
    blob_len = struct.unpack_from("<Q", out, 40)[0]
    assert blob_len == len(out) - 48 == 8 + 16 * 32 + 32
    assert struct.unpack_from("<Q", out, 48)[0] == 4  # the cap HEIGHT (write_usize), not its length
    with pytest.raises(mp2.Mp2gError):
        mp2.serialize_proof_with_vk(proof, cap[:12], dig)


def test_proof_with_vk_round_trip(mp2):
    """mp2g_proof_with_vk_deserialize is the inverse of serialize_proof + serialize_proof_with_vk (ProofWithVK::deserialize,
    mp2-common/src/proof.rs:54-57): the proof's parts, the key's cap and digest; a blob with another cap height, a truncated or a
    padded one is refused"""
    ofp, fp = make_fp(mp2)
    n = 1 << fp.log_n
    vals = [O.rand_field((w, n), 40 + i) for i, w in enumerate((5, 9, 4, 3))]
    cd, ph = O.rand_field(4, 1), O.rand_field(4, 2)
    caps, openings, proof = O.pcs_prove(ofp, vals, cd, ph)
    pis = O.rand_field(7, 3)
    blob = mp2.serialize_proof_with_vk(mp2.serialize_proof(fp, 2, caps, openings, proof, pis), caps[0].reshape(-1, 4), cd)
    (c2, o2, p2, pi2), vk_cap, dig = mp2.deserialize_proof_with_vk(fp, 2, blob, 7)
    assert np.array_equal(c2, caps) and np.array_equal(o2, openings) and np.array_equal(p2, proof) and np.array_equal(pi2, pis)
    assert np.array_equal(vk_cap.ravel(), caps[0]) and np.array_equal(dig, cd)
    assert O.pcs_verify(ofp, dig, ph, c2, o2, p2) == 0
    for bad in (blob[:-8], blob + bytes(8)):
        with pytest.raises(mp2.Mp2gError):
            mp2.deserialize_proof_with_vk(fp, 2, bad, 7)
    with pytest.raises(mp2.Mp2gError):
        mp2.deserialize_proof_with_vk(fp, 2, blob, 7, vk_cap_len=8)


def test_proof_store(tmp_path):
    """proofstore.py = the harness store of mp2-v1/tests/common/proof_storage.rs: store / get_proof_exact / move_proof by ProofKey;
    a second store over the same directory (another call) finds what the first one kept; a missing key is an error naming it"""
    PS = __import__("importlib").import_module("mapreduce-plonky2_amd.proofstore")
    st = PS.ProofStore(str(tmp_path / "s"))
    k1, k2 = PS.ProofKey.row("t", 1, "00ff"), PS.ProofKey.cell("t", 1, "00ff", 3)
    assert k1 != k2 and k1 == PS.ProofKey.row("t", 1, "00ff") and k1.compute_hash() != PS.ProofKey.row("t", 2, "00ff").compute_hash()
    st.store_proof(k1, b"abc", {"rows": 3})
    st.store_proof(k2, bytes(range(200)))
    st.store_proof(k1, b"abcd", {"rows": 4})  # the latest proof under a key counts
    again = PS.ProofStore(str(tmp_path / "s"))
    assert again.get_proof_exact(k1) == b"abcd" and again.note(k1) == {"rows": 4} and again.get_proof_exact(k2) == bytes(range(200))
    assert again.contains(k2) and not again.contains(PS.ProofKey.index("t", 9))
    with pytest.raises(KeyError, match="index_tree"):
        again.get_proof_exact(PS.ProofKey.index("t", 9))
    k3 = PS.ProofKey.row("t", 2, "00ff")
    again.move_proof(k1, k3)
    again.move_proof(PS.ProofKey.index("t", 9), k1)  # silent
    assert not again.contains(k1) and again.get_proof_exact(k3) == b"abcd" and len(again.keys()) == 2
    assert not [f for f in __import__("os").listdir(str(tmp_path / "s")) if f.endswith(".tmp")]
    # one file per key: the header travels with its proof (no orphan note, no note of one writer beside the bytes of another)
    import os
    assert sorted(f.rsplit(".", 1)[1] for f in os.listdir(str(tmp_path / "s"))) == ["bin", "bin"]
    # the header's clear-text key is compared on every read: a file under another key's hash does not alias it
    os.replace(again._file(k3), again._file(k1))
    with pytest.raises(KeyError, match="hash collision"):
        again.get_proof_exact(k1)
    again.remove(k1)
    assert not again.contains(k1) and again.keys() == [k2.canonical()]
    # a file cut short is refused, not returned as a shorter proof
    with open(again._file(k2), "r+b") as f:
        f.truncate(os.path.getsize(again._file(k2)) - 5)
    with pytest.raises(ValueError, match="proof bytes"):
        again.get_proof_exact(k2)
    # the committed store of round 5 (proof bytes in the .bin, header in a .json beside it) is still read
    old = PS.ProofStore(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05", "table_2p20_store"))
    ks = old.keys()
    assert len(ks) >= 8 and all(k.startswith("row_tree.") for k in ks)


def fnv1a(data):
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.mark.parametrize("log_n,ws,nq", [(6, (5, 9, 4, 3), 3), (12, (84, 135, 20, 16), 28)])
def test_wire_format_second_opinion(mp2, log_n, ws, nq):
    """mp2g_proof_serialize / mp2g_proof_with_vk_serialize against the independent bincode writer of
    tests/bincode_ref.py (written from plonky2's type definitions), on a proof of the standard 2^12 shape"""
    import bincode_ref as BR
    ofp = O.standard_params(log_n, ws, pow_bits=4, num_queries=nq)
    fp = mp2.FriParams()
    ctypes.memmove(ctypes.byref(fp), ctypes.byref(ofp), ctypes.sizeof(fp))
    n = 1 << log_n
    vals = [O.rand_field((w, n), 40 + i) for i, w in enumerate(ws)]
    cd, ph = O.rand_field(4, 1), O.rand_field(4, 2)
    caps, openings, proof = O.pcs_prove(ofp, vals, cd, ph)
    pis = O.rand_field(9, 3)
    num_constants = ws[0] - (80 if ws[0] > 80 else 3)
    got = mp2.serialize_proof(fp, num_constants, caps, openings, proof, pis)
    nested = BR.structured(fp, num_constants, caps, openings, proof, pis)
    want = BR.proof_with_public_inputs(nested)
    assert got == want
    assert len(nested["openings"]["lookup_zs"]) == 0 and len(nested["opening_proof"]["query_round_proofs"]) == nq
    vk_cap = caps[0].reshape(-1, 4)
    got_vk = mp2.serialize_proof_with_vk(got, vk_cap, cd)
    assert got_vk == BR.proof_with_vk(nested, [list(map(int, h)) for h in vk_cap], [int(x) for x in cd])
    if log_n == 12:  # the committed fingerprint of both serializers on the standard-shape proof
        import json, os
        k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wire_fnv.json")))
        assert {"proof_bytes": len(got), "proof_fnv1a": fnv1a(got), "proof_with_vk_fnv1a": fnv1a(got_vk)} == k["standard_2p12"]


def test_shard_ranges_cover(mp2):
    sh = __import__("importlib").import_module("mapreduce-plonky2_amd.sharding")
    for n, world in ((1 << 20, 8), (1024, 3), (5, 8), (0, 4)):
        got = [sh.shard_range(n, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == n
        assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
        for i in range(0, n, max(1, n // 97)):
            r = sh.owner_of(i, n, world)
            assert got[r][0] <= i < got[r][1]


def test_tree_handoff_plan(mp2):
    sh = __import__("importlib").import_module("mapreduce-plonky2_amd.sharding")
    # 1024 leaves, binary tree, 8 ranks: 1023 parents; only the top log2(8)=3 levels move proofs
    plan = sh.subtree_plan(1024, 2, 8)
    assert sum(len(l) for l in plan) == 1023
    moves = sh.tree_handoff_plan(1024, 2, 8)
    assert len(moves) == 4 + 2 + 1 and {m[0] for m in moves} == {7, 8, 9}
    # children before parents: every child index at level l was produced at level l-1
    for l, lvl in enumerate(plan[1:], 1):
        produced = {node for node, _, _ in plan[l - 1]}
        assert all(c in produced for _, _, ch in lvl for c in ch)
    assert sh.tree_handoff_plan(64, 2, 1) == []
