#!/usr/bin/env python3
"""Extract the known-answer DATA of the reference's DerivedData unit tests
(tests/unit/Mesh/TestDerivedData.cpp: genEsuf :2429, genInpofa :2767,
genGeoFaceTri :3096, genGeoElemTet :3160) into tests/golden/derived_data_ka.json.

Only the brace-initialised integer tables (inputs and expected outputs) are
read; the tiny tetrahedron geometry cases are transcribed by value.  Runs in
the development container only (needs /root/reference)."""
import json
import os
import re

SRC = "/root/reference/tests/unit/Mesh/TestDerivedData.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))


def table(lines, start, name):
    """integers of `name { ... };` starting at or after 1-based line `start`"""
    txt = "".join(lines[start - 1:])
    m = re.search(re.escape(name) + r"\s*\{(.*?)\};", txt, re.S)
    return [int(x) for x in re.findall(r"-?\d+", m.group(1))]


def main():
    lines = open(SRC).read().splitlines(keepends=True)
    out = {
        "genEsuf": {"inpoel_1based": table(lines, 2429, "inpoel"),
                    "belem_1based": table(lines, 2429, "belem"),
                    "nbfac": 48, "nipfac": 170,
                    "correct_esuf_1based": table(lines, 2429, "correct_esuf")},
        "genInpofa": {"inpoel_1based": table(lines, 2767, "inpoel"),
                      "triinpoel_1based": table(lines, 2767, "triinpoel"),
                      "nbfac": 48,
                      "correct_inpofa_1based": table(lines, 2767, "correct_inpofa")},
        "genGeoFaceTri": {"coord": [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]],
                          "inpofa": [0, 1, 2, 0, 3, 1, 1, 3, 2, 2, 3, 0],
                          "farea": [0.5, 0.5, 0.5, 0.8660254037844389],
                          "fnorm": [[0.0, 0.0, -1.0, 3.0 ** -0.5], [0.0, -1.0, 0.0, 3.0 ** -0.5],
                                    [-1.0, 0.0, 0.0, 3.0 ** -0.5]],
                          "fcent": [[1 / 3, 1 / 3, 0.0, 1 / 3], [1 / 3, 0.0, 1 / 3, 1 / 3],
                                    [0.0, 1 / 3, 1 / 3, 1 / 3]]},
        "genGeoElemTet": {"coord": [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]],
                          "inpoel": [0, 3, 2, 1], "vol": 1.0 / 6.0, "cent": [0.25, 0.25, 0.25]},
    }
    with open(os.path.join(HERE, "derived_data_ka.json"), "w") as fh:
        json.dump(out, fh)
    for k, v in out.items():
        print(k, {a: (len(b) if isinstance(b, list) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
