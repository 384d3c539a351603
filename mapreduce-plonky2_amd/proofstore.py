"""A proof store keyed by ProofKey: the host-side mirror of the reference's test harness store
(mp2-v1/tests/common/proof_storage.rs:58-140 `ProofKey`, `ProofStorage::{store_proof, get_proof_exact, move_proof}`, :211-274 the
key-value implementation). The reference keeps every proof of a table build in it so that a later step (a parent node, a later
block, another process) picks a child up as serialized `ProofWithVK` bytes (mp2-common/src/proof.rs:42-57); here it is what lets a
table too large for one GPU lease be built block by block across calls -- every block's root goes in as ProofWithVK bytes
(mp2g_proof_with_vk_serialize), the join levels take them out (mp2g_proof_with_vk_deserialize).

A directory of files, not a storage engine: one file per key (header + proof), written to a temporary name and renamed, so a call
that is cut off leaves either the whole proof with its header or nothing."""
import json
import os

# ProofKey variants of proof_storage.rs:58-79 that sit on the table-creation path
KINDS = ("cell_tree", "row_tree", "index_tree", "ivc")


class ProofKey:
    """proof_storage.rs:58-79. `kind` is the prefix the reference mixes into the key's hash (:93-140); the identifier fields are
    CellProofIdentifier {table, primary, secondary, tree_key} / RowProofIdentifier {table, primary, tree_key} /
    IndexProofIdentifier {table, tree_key} (:19-53)."""

    def __init__(self, kind, **ident):
        assert kind in KINDS, kind
        self.kind, self.ident = kind, dict(sorted(ident.items()))

    @classmethod
    def cell(cls, table, primary, secondary, tree_key):
        return cls("cell_tree", table=table, primary=int(primary), secondary=str(secondary), tree_key=int(tree_key))

    @classmethod
    def row(cls, table, primary, tree_key):
        return cls("row_tree", table=table, primary=int(primary), tree_key=str(tree_key))

    @classmethod
    def index(cls, table, tree_key):
        return cls("index_tree", table=table, tree_key=int(tree_key))

    def canonical(self):
        return self.kind + "." + ".".join(f"{k}={v}" for k, v in self.ident.items())

    def compute_hash(self):
        """proof_storage.rs:82-90 hashes the key with Rust's DefaultHasher; any stable 64-bit hash serves the store: FNV-1a"""
        h = 0xCBF29CE484222325
        for b in self.canonical().encode():
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h

    def __eq__(self, other):
        return isinstance(other, ProofKey) and self.canonical() == other.canonical()

    def __hash__(self):
        return hash(self.canonical())

    def __repr__(self):
        return f"ProofKey({self.canonical()})"


MAGIC = b"MP2GPRF1"


class ProofStore:
    """ProofStorage (proof_storage.rs:139-140) over a directory. ONE file per key, <kind>_<hash:016x>.bin = MAGIC, u32 header length,
    header (JSON: the key in clear, the proof's length, what the caller wants kept beside the proof), the proof bytes -- written to a
    temporary name and renamed, so that a key holds a whole proof WITH its own header or nothing: a call that is cut off, or two ranks
    storing under one key, cannot leave the note of one proof beside the bytes of another. The header's clear-text key is compared with
    the key asked for on every read (two keys whose 64-bit hashes collide cannot alias). The note is never read back for the proof
    itself: a loaded proof is checked against its verifier key and public inputs. Stores written before round 6 (the proof bytes alone
    in the .bin, the header in a .json beside it: profiles/r05/table_2p20_store) are still read."""

    def __init__(self, path):
        self.path = path
        os.makedirs(path, exist_ok=True)

    def _file(self, key, ext="bin"):
        return os.path.join(self.path, f"{key.kind}_{key.compute_hash():016x}.{ext}")

    def _write(self, name, data):
        # a temporary name of this writer's own: two ranks storing under one key must not share it (the later rename wins, whole)
        import threading
        tmp = f"{name}.{os.getpid()}.{threading.get_ident()}.tmp"
        with open(tmp, "wb") as f:
            f.write(data)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, name)

    def _read(self, path):
        """(header dict or None, proof bytes) of a store file: the one-file form, or the proof alone of an older store"""
        with open(path, "rb") as f:
            data = f.read()
        if data[:8] == MAGIC:
            n = int.from_bytes(data[8:12], "little")
            head = json.loads(data[12:12 + n].decode())
            proof = data[12 + n:]
            if len(proof) != head["bytes"]:
                raise ValueError(f"{path}: the header announces {head['bytes']} proof bytes, the file holds {len(proof)}")
            return head, proof
        side = path[:-4] + ".json"
        head = None
        if os.path.exists(side):
            with open(side) as f:
                head = json.load(f)
        return head, data

    def store_proof(self, key, proof, note=None):
        """store_proof(key, proof): overwrites an earlier proof under the same key (proof_storage.rs:30-33: the latest one counts)"""
        head = json.dumps({"key": key.canonical(), "bytes": len(proof), "note": note or {}}).encode()
        self._write(self._file(key), MAGIC + len(head).to_bytes(4, "little") + head + bytes(proof))
        try:
            os.remove(self._file(key, "json"))  # the side file of an older store's entry under this key
        except FileNotFoundError:
            pass

    def contains(self, key):
        return os.path.exists(self._file(key))

    def _entry(self, key):
        try:
            head, proof = self._read(self._file(key))
        except FileNotFoundError:
            raise KeyError(f"proof with key {key!r} not found in {self.path}") from None
        if head is not None and head.get("key") != key.canonical():
            raise KeyError(f"proof with key {key!r} not found in {self.path}: the file of its hash holds the key {head.get('key')!r} (a 64-bit hash collision)")
        return head, proof

    def get_proof_exact(self, key):
        """get_proof_exact(key): the bytes, or KeyError naming the key (proof_storage.rs:262-272 `proof with key .. not found`)"""
        return self._entry(key)[1]

    def note(self, key):
        head, _ = self._entry(key)
        if head is None:
            raise KeyError(f"proof with key {key!r} has no header in {self.path}")
        return head["note"]

    def remove(self, key):
        """drop a key's proof (a stored proof that failed its checks: the caller rebuilds it); silent when there is none"""
        for ext in ("bin", "json"):
            try:
                os.remove(self._file(key, ext))
            except FileNotFoundError:
                pass

    def move_proof(self, old_key, new_key):
        """move_proof(old, new): a silent no-op when the old key holds nothing (proof_storage.rs:236-260)"""
        if not self.contains(old_key):
            return
        proof, note = self.get_proof_exact(old_key), self.note(old_key)
        self.store_proof(new_key, proof, note)
        self.remove(old_key)

    def keys(self):
        """the clear-text keys of the proofs the store holds (a header without its proof does not exist in the one-file form; an
        older store's side file without a .bin is not a proof)"""
        out = []
        for name in sorted(os.listdir(self.path)):
            if name.endswith(".bin"):
                head, _ = self._read(os.path.join(self.path, name))
                if head is not None:
                    out.append(head["key"])
        return out
