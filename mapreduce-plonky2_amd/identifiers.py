"""Column identifiers of mp2-v1 (off-circuit `Hasher::hash_no_pad` call sites of SURVEY 8 row a5), mirrored over the
C ABI: mp2-v1/src/values_extraction/mod.rs:41-62,157-296. Every identifier is limb 0 of hash_no_pad over a byte
string, one field element per byte (`impl ToFields for &[u8]`, mp2-common/src/utils.rs:390-394). Batch them: one
`hash_no_pad_batch` call per group of equal length."""
import numpy as np

from . import POSEIDON2

KEY_ID_PREFIX = b"\0KEY"
INNER_KEY_ID_PREFIX = b"\0\0IN_KEY"
OUTER_KEY_ID_PREFIX = b"\0OUT_KEY"
BLOCK_ID_DST = b"BLOCK_NUMBER"
OFFCHAIN_TABLE_DST = b"OFFCHAIN_TABLE"


def identifiers_batch(ctx, byte_strings, variant=POSEIDON2):
    """limb 0 of hash_no_pad over every byte string (one field element per byte): one `hash_no_pad_batch` launch per
    group of strings of equal length, in the callers' order. The identifiers of a whole table (or of every table of
    a block) go through here together."""
    out = [0] * len(byte_strings)
    by_len = {}
    for i, b in enumerate(byte_strings):
        by_len.setdefault(len(b), []).append(i)
    for ln, idxs in by_len.items():
        limbs = np.frombuffer(b"".join(bytes(byte_strings[i]) for i in idxs), dtype=np.uint8).astype(np.uint64).reshape(len(idxs), ln)
        h = ctx.hash_no_pad_batch(limbs, 4, variant)
        for i, row in zip(idxs, h):
            out[i] = int(row[0])
    return out


def _id(ctx, data, variant):
    return identifiers_batch(ctx, [data], variant)[0]


def value_column_preimage(slot, byte_offset, length, evm_word, extra):
    """the byte string identifier_for_value_column_raw hashes (mod.rs:185-196)"""
    return bytes([slot]) + int(byte_offset).to_bytes(8, "big") + int(length).to_bytes(8, "big") + int(evm_word).to_bytes(4, "big") + bytes(extra)


def table_column_identifiers(ctx, slot_inputs, contract_address, chain_id, extra=b"", variant=POSEIDON2):
    """identifier_for_value_column of every (slot, byte_offset, length, evm_word) of a table in one batch
    (TableMetadata's extracted columns, mod.rs:95-150)"""
    ex = identifier_raw_extra(contract_address, chain_id, extra)
    return identifiers_batch(ctx, [value_column_preimage(s, o, ln, w, ex) for (s, o, ln, w) in slot_inputs], variant)


def identifier_raw_extra(contract_address, chain_id, extra=b""):
    """mod.rs:273-280: contract_address (20 bytes) || chain_id (u64, big endian) || extra"""
    assert len(contract_address) == 20
    return bytes(contract_address) + int(chain_id).to_bytes(8, "big") + bytes(extra)


def identifier_block_column(ctx, variant=POSEIDON2):
    """mod.rs:157-160"""
    return _id(ctx, BLOCK_ID_DST, variant)


def identifier_offchain_column(ctx, table_name, column_name, variant=POSEIDON2):
    """mod.rs:52-59"""
    return _id(ctx, OFFCHAIN_TABLE_DST + table_name.encode() + column_name.encode(), variant)


def identifier_for_value_column_raw(ctx, slot, byte_offset, length, evm_word, extra, variant=POSEIDON2):
    """mod.rs:185-196: H(slot || byte_offset (usize BE) || length (usize BE) || evm_word (u32 BE) || extra)[0]"""
    return _id(ctx, value_column_preimage(slot, byte_offset, length, evm_word, extra), variant)


def identifier_for_value_column(ctx, slot, byte_offset, length, evm_word, contract_address, chain_id, extra=b"", variant=POSEIDON2):
    """mod.rs:166-180"""
    return identifier_for_value_column_raw(ctx, slot, byte_offset, length, evm_word, identifier_raw_extra(contract_address, chain_id, extra), variant)


def _with_prefix(ctx, prefix, slot, contract_address, chain_id, extra, variant):
    """mod.rs:260-296 compute_id_with_prefix(_raw): H(prefix || slot || contract_address || chain_id || extra)[0]"""
    return _id(ctx, prefix + bytes([slot]) + identifier_raw_extra(contract_address, chain_id, extra), variant)


def identifier_for_mapping_key_column(ctx, slot, contract_address, chain_id, extra=b"", variant=POSEIDON2):
    return _with_prefix(ctx, KEY_ID_PREFIX, slot, contract_address, chain_id, extra, variant)


def identifier_for_outer_mapping_key_column(ctx, slot, contract_address, chain_id, extra=b"", variant=POSEIDON2):
    return _with_prefix(ctx, OUTER_KEY_ID_PREFIX, slot, contract_address, chain_id, extra, variant)


def identifier_for_inner_mapping_key_column(ctx, slot, contract_address, chain_id, extra=b"", variant=POSEIDON2):
    return _with_prefix(ctx, INNER_KEY_ID_PREFIX, slot, contract_address, chain_id, extra, variant)
