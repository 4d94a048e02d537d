#!/usr/bin/env python
"""Merge tune-table entries (files written by tools/split_sweep.py --emit-table) into the committed table: an entry replaces the
committed entry of the same key; when the new configuration is of a split-operand family (5, 6) and the old one is not, the old
entry goes to wav2lip_amd/tune_table_nosplit.json (what W2L_SPLIT=0 puts back) unless that file already holds the key.

    python tools/merge_tune_entries.py gpurun_out/x/table_128.json [more.json ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(path, doc):
    with open(path, "w") as fh:
        fh.write(json.dumps(doc, indent=None, separators=(",", ":")).replace("],[", "],\n[") + "\n")


def main():
    main_p = os.path.join(ROOT, "wav2lip_amd", "tune_table.json")
    nos_p = os.path.join(ROOT, "wav2lip_amd", "tune_table_nosplit.json")
    main_d, nos_d = json.load(open(main_p)), json.load(open(nos_p))
    nk = main_d["key_ints"]
    table = {tuple(e[:nk]): e[nk:] for e in main_d["entries"]}
    nosplit = {tuple(e[:nk]): e[nk:] for e in nos_d["entries"]}
    nconf = main_d["num_configs"]
    # family of an id without the library: ids [13, 18] family 5, id 19 family 6 (conv_igemm.hip: appended, never renumbered)
    split_family = lambda c: c >= 13      # noqa: E731
    changed = 0
    for path in sys.argv[1:]:
        doc = json.load(open(path))
        assert doc["key_ints"] == nk
        nconf = max(nconf, doc["num_configs"])
        for e in doc["entries"]:
            key, val = tuple(e[:nk]), e[nk:]
            old = table.get(key)
            if old == val:
                continue
            if old is not None and split_family(val[0]) and not split_family(old[0]) and key not in nosplit:
                nosplit[key] = old
            table[key] = val
            changed += 1
            print("%s: %s -> %s" % (key, old, val))
    main_d["entries"] = sorted(list(k) + v for k, v in table.items())
    main_d["num_configs"] = nconf
    nos_d["entries"] = sorted(list(k) + v for k, v in nosplit.items())
    nos_d["num_configs"] = nconf
    dump(main_p, main_d)
    dump(nos_p, nos_d)
    print("%d entries changed; table %d entries, nosplit %d entries" % (changed, len(table), len(nosplit)))


if __name__ == "__main__":
    main()
