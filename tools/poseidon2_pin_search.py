#!/usr/bin/env python3
"""Search for a hash formula that reproduces the column identifiers stored in the reference tree -- the only
candidate known-answer values for the Poseidon2 sponge.

Targets:
  * mp2-v1/tests/integrated_tests.rs:308   block_number id 15542555334667826467 (+ four value-column ids whose
    contract address is not in the tree)
  * parsil/tests/context.json:88-98        block_number 17422912802427138938 (already pinned: Poseidon,
    BE-u32-packed b"BLOCK_NUMBER"), map_value 12191544657365810443, map_key 10362498354857054631

Matrix: {Poseidon, Poseidon2 (HorizenLabs instance as restated in oracle/), Poseidon2 without the initial linear
layer, Poseidon2 with Plonky3's M4, Poseidon2 with round constants added after the S-box layer}
x {hash_no_pad, hash_pad} x {1 byte per limb, BE u32, LE u32, BE u64, LE u64, BE u32 left-padded}
x output limb {0..3} x a list of plausible domain-separation strings; and, for the mapping ids, the formulas of
mp2-v1/src/values_extraction/mod.rs:166-296 and their older shapes over slots 0..15, the anvil default deployer's
first six contract addresses and chain id 31337.

Writes every hit (or "no hit") to stdout; profiles/r02/poseidon2_pin_search.txt is the committed run.
"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib  # noqa: E402

C = importlib.import_module("mapreduce-plonky2_amd.circuits")
P = C.P
K = C.poseidon2_constants()

TARGETS = {
    15542555334667826467: "integrated_tests.rs:308 block_number",
    10143644063834010325: "integrated_tests.rs:308 field1", 14738928498191419754: "integrated_tests.rs:308 field2",
    2724380514203373020: "integrated_tests.rs:308 field3", 1084192582840933701: "integrated_tests.rs:308 field4",
    17422912802427138938: "context.json:88 block_number", 12191544657365810443: "context.json:93 map_value",
    10362498354857054631: "context.json:98 map_key",
}
M4_HL = ((5, 7, 1, 3), (4, 6, 1, 1), (1, 3, 5, 7), (1, 1, 4, 6))
M4_P3 = ((2, 3, 1, 1), (1, 2, 3, 1), (1, 1, 2, 3), (3, 1, 1, 2))


def ext_layer(s, m4):
    t = [sum(s[4 * c + j] * m4[i][j] for j in range(4)) % P for c in range(3) for i in range(4)]
    sums = [(t[i] + t[4 + i] + t[8 + i]) % P for i in range(4)]
    return [(t[4 * c + i] + sums[i]) % P for c in range(3) for i in range(4)]


def int_layer(s):
    d = K["POSEIDON2_DIAG_M1"]
    tot = sum(s) % P
    return [(s[i] * d[i] + tot) % P for i in range(12)]


def poseidon2(s, m4=M4_HL, initial=True, rc_after=False):
    s = list(s)
    if initial:
        s = ext_layer(s, m4)
    rce, rci = K["POSEIDON2_RC_EXT"], K["POSEIDON2_RC_INT"]
    for r in range(8):
        if r == 4:
            for q in range(22):
                if rc_after:
                    s[0] = (pow(s[0], 7, P) + rci[q]) % P
                else:
                    s[0] = pow((s[0] + rci[q]) % P, 7, P)
                s = int_layer(s)
        if rc_after:
            s = [(pow(x, 7, P) + rce[12 * r + i]) % P for i, x in enumerate(s)]
        else:
            s = [pow((x + rce[12 * r + i]) % P, 7, P) for i, x in enumerate(s)]
        s = ext_layer(s, m4)
    return s


def poseidon(s):
    s = list(s)
    rc = K["POSEIDON_RC"]
    for r in range(30):
        s = [(s[i] + rc[12 * r + i]) % P for i in range(12)]
        if 4 <= r < 26:
            s[0] = pow(s[0], 7, P)
        else:
            s = [pow(x, 7, P) for x in s]
        s = C.poseidon_mds(s)
    return s


PERMS = {"poseidon": poseidon, "poseidon2": poseidon2, "poseidon2/no-initial-layer": lambda s: poseidon2(s, initial=False),
         "poseidon2/plonky3-M4": lambda s: poseidon2(s, m4=M4_P3), "poseidon2/rc-after-sbox": lambda s: poseidon2(s, rc_after=True),
         "poseidon2/plonky3-M4/no-initial": lambda s: poseidon2(s, m4=M4_P3, initial=False)}


def sponge(perm, inputs, pad):
    v = list(inputs)
    if pad:  # hash_pad: 1, zeros to rate - 1, 1
        v.append(1)
        while (len(v) + 1) % 8:
            v.append(0)
        v.append(1)
    st = [0] * 12
    for i in range(0, len(v), 8):
        chunk = v[i:i + 8]
        st[:len(chunk)] = chunk
        st = perm(st)
    if not v:
        st = perm(st)
    return st[:4]


def packings(b):
    out = {"byte/limb": list(b)}
    for w, name in ((4, "u32"), (8, "u64")):
        r = b + bytes((-len(b)) % w)
        l = bytes((-len(b)) % w) + b
        out[f"BE {name} right-padded"] = [int.from_bytes(r[i:i + w], "big") % P for i in range(0, len(r), w)]
        out[f"LE {name} right-padded"] = [int.from_bytes(r[i:i + w], "little") % P for i in range(0, len(r), w)]
        out[f"BE {name} left-padded"] = [int.from_bytes(l[i:i + w], "big") % P for i in range(0, len(l), w)]
    return out


def try_bytes(label, b, hits):
    n = 0
    for pname, perm in PERMS.items():
        for pk, limbs in packings(b).items():
            for pad in (False, True):
                out = sponge(perm, limbs, pad)
                n += 1
                for k, v in enumerate(out):
                    if v in TARGETS:
                        hits.append(f"HIT {TARGETS[v]} = {pname} {'hash_pad' if pad else 'hash_no_pad'}({label}, {pk})[{k}]")
    return n


def main():
    hits, tried = [], 0
    dsts = [b"BLOCK_NUMBER", b"block_number", b"BLOCK", b"block", b"BLOCK_ID", b"BLOCK_NUM", b"blocknumber", b"BlockNumber", b"blockNumber",
            b"BLOCKNUMBER", b"BLOCK_NUMBER\0", b"\0BLOCK_NUMBER", b"block_number_column", b"BLOCK_NUMBER_ID", b"PRIMARY_INDEX", b"primary_index",
            b"BLOCK_ID_DST", b"block_id", b"number", b"NUMBER"]
    for d in dsts:
        tried += try_bytes(repr(d), d, hits)
    # mapping table of the integration tests: slot = MAPPING_SLOT 4 (mp2-v1/tests/common/cases/indexing.rs:56); the anvil default
    # deployer's first contract addresses; chain id 31337
    addrs = ["5FbDB2315678afecb6367f032d93F642f64180aa", "e7f1725E7734CE288F8367e1Bb143E90bb3F0512", "9fE46736679d2D9a65F0992F2272dE9f3c7fa6e0",
             "Cf7Ed3AccA5a467e9e704C703E8D87F634fB0FC9", "Dc64a140Aa3E981100a9becA4E685f962f0cF6C9", "5FC8d32690cc91D4c39d9d3abcBD16989F875707"]
    chain = (31337).to_bytes(8, "big")
    for a, slot in itertools.product(addrs, range(0, 12)):
        ab = bytes.fromhex(a)
        forms = {
            "KEY||slot||addr||chain": b"\0KEY" + bytes([slot]) + ab + chain, "KEY3||slot||addr||chain": b"KEY" + bytes([slot]) + ab + chain,
            "VAL||slot||addr||chain": b"\0VAL" + bytes([slot]) + ab + chain, "VAL3||slot||addr||chain": b"VAL" + bytes([slot]) + ab + chain,
            "slot||addr||chain": bytes([slot]) + ab + chain,
            "slot||off||len||word||addr||chain": bytes([slot]) + (0).to_bytes(8, "big") + (256).to_bytes(8, "big") + (0).to_bytes(4, "big") + ab + chain,
            "slot||off4||len4||word||addr||chain": bytes([slot]) + (0).to_bytes(4, "big") + (256).to_bytes(4, "big") + (0).to_bytes(4, "big") + ab + chain,
            "slot||off||len32||word||addr||chain": bytes([slot]) + (0).to_bytes(8, "big") + (32).to_bytes(8, "big") + (0).to_bytes(4, "big") + ab + chain,
        }
        for name, b in forms.items():
            tried += try_bytes(f"{name} slot={slot} addr=0x{a}", b, hits)
    print(f"{tried} sponge evaluations x 4 output limbs against {len(TARGETS)} stored identifiers")
    if hits:
        print("\n".join(sorted(set(hits))))
    only_known = all("context.json:88" in h for h in hits)
    print("no new hit: the Poseidon2 sponge stays unpinned against the reference" if only_known else "NEW HIT(S) ABOVE")


if __name__ == "__main__":
    main()
