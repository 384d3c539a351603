"""Independent bincode writer for plonky2's ProofWithPublicInputs and the reference's ProofWithVK
(mp2-common/src/proof.rs:42-57,84-98) -- TEST INFRASTRUCTURE: the second opinion on csrc/wire.hip.

Written from the type definitions, not from the flat layout walker of the product: every struct of the serde
tree is a function here, field by field in declaration order (plonky2 plonk/proof.rs, fri/proof.rs,
hash/merkle_tree.rs, hash/merkle_proofs.rs). bincode 1.3 defaults: fixed-width little-endian integers, a u64
length before every Vec / byte string, nothing before arrays, tuples, structs and newtypes.
"""
import struct


def u64(x):
    return struct.pack("<Q", int(x))


def vec(items, enc):
    return u64(len(items)) + b"".join(enc(i) for i in items)


def field(x):  # GoldilocksField(u64): newtype struct -> the u64
    assert 0 <= int(x) < 0xFFFFFFFF00000001, "canonical field elements only"
    return u64(x)


def ext(x):  # QuadraticExtension([F; 2]): newtype over an array -> the two limbs
    return field(x[0]) + field(x[1])


def hash_out(h):  # HashOut { elements: [F; 4] }
    return b"".join(field(e) for e in h)


def merkle_cap(cap):  # MerkleCap(Vec<HashOut>)
    return vec(cap, hash_out)


def merkle_proof(siblings):  # MerkleProof { siblings: Vec<HashOut> }
    return vec(siblings, hash_out)


def opening_set(o):
    """OpeningSet { constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys,
    lookup_zs, lookup_zs_next }: nine Vec<Extension>"""
    return b"".join(vec(o[k], ext) for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products",
                                               "quotient_polys", "lookup_zs", "lookup_zs_next"))


def fri_query_round(q):
    """FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs: Vec<(Vec<F>, MerkleProof)> },
    steps: Vec<FriQueryStep { evals: Vec<Extension>, merkle_proof }> }"""
    out = vec(q["evals_proofs"], lambda ep: vec(ep[0], field) + merkle_proof(ep[1]))
    return out + vec(q["steps"], lambda st: vec(st[0], ext) + merkle_proof(st[1]))


def fri_proof(f):
    """FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly: PolynomialCoeffs { coeffs }, pow_witness }"""
    return vec(f["commit_phase_merkle_caps"], merkle_cap) + vec(f["query_round_proofs"], fri_query_round) + vec(f["final_poly"], ext) + \
        field(f["pow_witness"])


def proof_with_public_inputs(p):
    """ProofWithPublicInputs { proof: Proof { wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap, openings,
    opening_proof }, public_inputs: Vec<F> }"""
    return merkle_cap(p["wires_cap"]) + merkle_cap(p["plonk_zs_partial_products_cap"]) + merkle_cap(p["quotient_polys_cap"]) + \
        opening_set(p["openings"]) + fri_proof(p["opening_proof"]) + vec(p["public_inputs"], field)


def verifier_only_to_bytes(cap, circuit_digest):
    """[dep] plonky2 VerifierOnlyCircuitData::to_bytes (util/serialization write_verifier_only_circuit_data): the cap
    HEIGHT as a u64 (write_usize), the cap's hashes without a length, the circuit digest"""
    height = len(cap).bit_length() - 1
    assert 1 << height == len(cap)
    return u64(height) + b"".join(hash_out(h) for h in cap) + hash_out(circuit_digest)


def proof_with_vk(p, cap, circuit_digest):
    """ProofWithVK { proof, #[serde(serialize_with = serialize)] vk }: the vk goes through serialize_bytes, i.e. a
    u64 length and the to_bytes() blob (mp2-common/src/serialization/mod.rs:47-57)"""
    blob = verifier_only_to_bytes(cap, circuit_digest)
    return proof_with_public_inputs(p) + u64(len(blob)) + blob


def structured(fp, num_constants, caps, openings, fri, public_inputs, n_lookup=0):
    """The prover's flat outputs (include/mp2g.h layout) as the nested proof of plonky2's types."""
    capn = 1 << fp.cap_height
    cap = lambda words: [list(map(int, words[4 * i:4 * i + 4])) for i in range(len(words) // 4)]
    ws = [fp.oracle_w[o] for o in range(fp.n_oracles)]
    op = [tuple(map(int, e)) for e in openings]
    # flat openings = FRI batch order: oracle 0, wires, Z + partial products, quotient, lookup | Z next, lookup next
    o1, o2 = ws[0], ws[0] + ws[1]
    o3 = o2 + ws[2] - n_lookup
    n_zeta = sum(ws)
    zs = fp.zs_count
    n_pp = ws[2] - zs - n_lookup
    opening = {"constants": op[:num_constants], "plonk_sigmas": op[num_constants:o1], "wires": op[o1:o2], "plonk_zs": op[o2:o2 + zs],
               "plonk_zs_next": op[n_zeta:n_zeta + zs], "partial_products": op[o2 + zs:o2 + zs + n_pp], "quotient_polys": op[o3:o3 + ws[3]],
               "lookup_zs": op[o3 + ws[3]:o3 + ws[3] + n_lookup], "lookup_zs_next": op[n_zeta + zs:n_zeta + zs + n_lookup]}
    lg = fp.log_n + fp.rate_bits
    depth = lg - fp.cap_height
    pos = 0
    fri = [int(x) for x in fri]

    def take(k):
        nonlocal pos
        out = fri[pos:pos + k]
        pos += k
        return out
    commit_caps = [cap(take(4 * capn)) for _ in range(fp.n_layers)]
    rounds = []
    for _ in range(fp.num_queries):
        eps = []
        for o in range(fp.n_oracles):
            leaf = take(ws[o])
            eps.append((leaf, cap(take(4 * depth))))
        steps, clg = [], lg
        for i in range(fp.n_layers):
            a = 1 << fp.arity_bits[i]
            clg -= fp.arity_bits[i]
            ev = take(2 * a)
            steps.append(([(ev[2 * j], ev[2 * j + 1]) for j in range(a)], cap(take(4 * (clg - fp.cap_height)))))
        rounds.append({"evals_proofs": eps, "steps": steps})
    deg = fp.log_n - sum(fp.arity_bits[i] for i in range(fp.n_layers))
    fin = take(2 << deg)
    final_poly = [(fin[2 * j], fin[2 * j + 1]) for j in range(1 << deg)]
    pow_witness = take(1)[0]
    assert pos == len(fri)
    return {"wires_cap": cap(caps[1]), "plonk_zs_partial_products_cap": cap(caps[2]), "quotient_polys_cap": cap(caps[3]), "openings": opening,
            "opening_proof": {"commit_phase_merkle_caps": commit_caps, "query_round_proofs": rounds, "final_poly": final_poly,
                              "pow_witness": pow_witness},
            "public_inputs": [int(x) for x in public_inputs]}
