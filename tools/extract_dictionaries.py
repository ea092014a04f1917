#!/usr/bin/env python3
"""Regenerate aruco3_amd/data/dictionaries.{bin,json} from the reference crate.

The marker dictionaries are *data* (the public ArUco / AprilTag / ARTag / ARToolKit+
/ Chilitags code lists).  The reference keeps them as `&[u64]` literals in
`src/dictionaries.rs:5-19` plus a `{num_bits, tau}` record per name in `:30-113`.
This script parses those numbers and stores them as one little-endian u64 blob and a
small JSON index, so that no reference source text lives in this repository.

Run only where /root/reference exists (the build container); the outputs are
committed.
"""
import json
import re
import struct
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/dictionaries.rs")
OUT = Path(__file__).resolve().parent.parent / "aruco3_amd" / "data"


def main() -> None:
    text = REF.read_text()
    tables = {}
    alias = {}
    for m in re.finditer(r"const\s+(\w+)\s*:\s*&'static\s*\[u64\]\s*=\s*(&\[[^\]]*\]|\w+)\s*;", text):
        name, body = m.group(1), m.group(2)
        if body.startswith("&["):
            tables[name] = [int(tok, 16) for tok in re.findall(r"0x[0-9a-fA-F]+", body)]
        else:
            alias[name] = body
    for k, v in alias.items():
        tables[k] = tables[v]

    # phf map entries:  "NAME" => ARDictionary { num_bits: N, tau: T, code_list: IDENT }
    # (commented-out entries are skipped by stripping /* ... */ first)
    body = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    entries = []
    for m in re.finditer(
        r'"(\w+)"\s*=>\s*ARDictionary\s*\{\s*num_bits\s*:\s*(\d+)\s*,\s*tau\s*:\s*(\d+)\s*,\s*code_list\s*:\s*(\w+)\s*,?\s*\}',
        body,
    ):
        entries.append((m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)))
    assert len(entries) == 15, len(entries)  # 15 names over 14 distinct tables

    blob = bytearray()
    index = {}
    offsets = {}
    for name, num_bits, tau, ident in sorted(entries):
        codes = tables[ident]
        key = ident if ident not in alias else alias[ident]
        if key not in offsets:
            offsets[key] = len(blob) // 8
            blob += struct.pack("<%dQ" % len(codes), *codes)
        index[name] = {"num_bits": num_bits, "tau": tau, "offset": offsets[key], "count": len(codes)}
    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "dictionaries.bin").write_bytes(bytes(blob))
    (OUT / "dictionaries.json").write_text(json.dumps(index, indent=1, sort_keys=True) + "\n")
    for name, e in sorted(index.items()):
        print(f"{name:18s} count={e['count']:5d} bits={e['num_bits']:2d} tau={e['tau']:2d}")


if __name__ == "__main__":
    main()
