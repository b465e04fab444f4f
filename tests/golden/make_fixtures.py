#!/usr/bin/env python3
"""Convert the reference's own DG regression fixtures into small .npz vectors.

Runs ONLY in the development container (needs /root/reference). It reads
*data files* held by the reference's regression tests -- ExodusII meshes
(NetCDF CDF-2, readable with scipy), the committed golden ExodusII outputs
(`*.std.exo`) and the committed golden diagnostics tables (`diag*.std`) -- and
stores them as compressed numpy archives under tests/golden/.  No reference
source code is read, executed or copied by this script.

Source directory: tests/regression/inciter/compflow/Euler/** (see CASES).
Control-file parameters of each case (scheme, dt/cfl, nstep, flux, limiter,
BC side sets, material gamma, problem constants) are transcribed in
tests/golden/cases.json from the `.q` files named there.

Usage:  python tests/golden/make_fixtures.py [case ...]     (no argument: every case of cases.json)
"""
import json
import os
import sys

import numpy as np
from scipy.io import netcdf_file

REF = "/root/reference/tests/regression/inciter"
HERE = os.path.dirname(os.path.abspath(__file__))


def _str(chararr):
    return b"".join(chararr).decode().strip("\x00 ").strip()


def read_mesh(path):
    """ExodusII input mesh -> coord[nnode,3], inpoel[ne,4] (0-based),
    side sets as {id: triangles[n,3]} (node ids, order-independent keys)."""
    f = netcdf_file(path, "r", mmap=False)
    v = f.variables
    coord = np.stack([v["coordx"][:], v["coordy"][:], v["coordz"][:]], axis=1)
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    tri = None
    tet = None
    tri_first_id = None
    eid = 1
    nblk = f.dimensions["num_el_blk"]
    for b in range(1, nblk + 1):
        c = v["connect%d" % b]
        et = c.elem_type.decode().upper()
        arr = np.array(c[:], dtype=np.int64) - 1
        if et.startswith("TRI"):
            tri = arr
            tri_first_id = eid
        elif et.startswith("TET"):
            tet = arr
        eid += arr.shape[0]
    assert tet is not None
    ss = {}
    if "ss_prop1" in v:
        ids = np.array(v["ss_prop1"][:], dtype=np.int64)
        for k, sid in enumerate(ids, start=1):
            el = np.array(v["elem_ss%d" % k][:], dtype=np.int64)
            if tri is None:
                # side sets given as (tet, ExodusII side number): side 1..4 of a TETRA are its
                # nodes (1,2,4), (2,3,4), (1,4,3), (1,3,2)
                side = np.array(v["side_ss%d" % k][:], dtype=np.int64)
                tab = np.array([[0, 1, 3], [1, 2, 3], [0, 3, 2], [0, 2, 1]])
                ss[int(sid)] = tet[el - 1][np.arange(len(el))[:, None], tab[side - 1]]
                continue
            # every side set of the CompFlow fixtures references TRI-block
            # elements (file ids tri_first_id .. tri_first_id+ntri-1)
            loc = el - tri_first_id
            assert tri is not None and loc.min() >= 0 and loc.max() < len(tri)
            ss[int(sid)] = tri[loc]
    f.close()
    return coord, tet, ss


def read_golden_exo(path, ntet):
    """Golden ExodusII output -> times[nt], names[nv], vals[nt,nv,ne] for the
    TETRA block, plus the output mesh (to verify ordering is the input's)."""
    f = netcdf_file(path, "r", mmap=False)
    v = f.variables
    times = np.array(v["time_whole"][:], dtype=np.float64)
    names = [_str(r) for r in v["name_elem_var"][:]]
    blk = None
    for b in range(1, f.dimensions["num_el_blk"] + 1):
        if v["connect%d" % b].elem_type.decode().upper().startswith("TET"):
            blk = b
    conn = np.array(v["connect%d" % blk][:], dtype=np.int64) - 1
    assert conn.shape[0] == ntet
    vals = np.zeros((len(times), len(names), ntet))
    for i in range(len(names)):
        vals[:, i, :] = v["vals_elem_var%deb%d" % (i + 1, blk)][:]
    coord = np.stack([v["coordx"][:], v["coordy"][:], v["coordz"][:]], axis=1)
    f.close()
    return times, names, vals, conn, np.array(coord, dtype=np.float64)


def read_diag(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            if line.lstrip().startswith("#") or not line.strip():
                continue
            rows.append([float(x) for x in line.split()])
    return np.array(rows)


def make_udu():
    """The reference's t0 mesh sequence uniform -> uniform_derefine -> uniform (-> uniform_derefine -> uniform):
    the meshes its regression test amr_t0ref_ud(ud)u_trans_dg compares after every initial refinement step
    (mesh_refinement/t0ref/gauss_hump_dg_uniform_deref_t0ref.std.e-s.{0..5}.1.0, CMakeLists.txt:108-160) ->
    tests/golden/t0ref_gauss_hump_udu.npz: s{k}_coord, s{k}_inpoel, s{k}_ss_ids, s{k}_ss_tri_{id}.  Pins the uniform
    derefinement (qdg_derefine_uniform, qdg_mesh_derefine_uniform): stage 2 is the initial mesh again."""
    d = os.path.join(REF, "mesh_refinement", "t0ref")
    out = {}
    for k in range(6):
        coord, tet, ss = read_mesh(os.path.join(d, "gauss_hump_dg_uniform_deref_t0ref.std.e-s.%d.1.0" % k))
        out["s%d_coord" % k] = coord
        out["s%d_inpoel" % k] = tet.astype(np.int32)
        out["s%d_ss_ids" % k] = np.array(sorted(ss), dtype=np.int64)
        for sid, tri in ss.items():
            out["s%d_ss_tri_%d" % (k, sid)] = tri.astype(np.int32)
    path = os.path.join(HERE, "t0ref_gauss_hump_udu.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "udu":
        make_udu()
        return
    with open(os.path.join(HERE, "cases.json")) as fh:
        cases = json.load(fh)
    only = set(sys.argv[1:])
    for name, c in cases.items():
        if only and name not in only:
            continue
        d = os.path.join(REF, c["dir"])
        # (mesh_is_golden_output: a run with initial mesh refinement computes on the mesh the reference's
        # Refiner produced; that mesh -- coordinates, tets, side sets -- is stored in the golden output file)
        coord, inpoel, ss = read_mesh(os.path.join(d, c["mesh"]))
        out = {"coord": coord, "inpoel": inpoel.astype(np.int64),
               "ss_ids": np.array(sorted(ss), dtype=np.int64)}
        for sid in ss:
            out["ss_tri_%d" % sid] = ss[sid].astype(np.int64)
        if c.get("input_mesh"):
            # the unrefined input mesh of that run, to check this repository's uniform refinement against
            ic, ii, iss = read_mesh(os.path.join(d, c["input_mesh"]))
            out["in_coord"], out["in_inpoel"] = ic, ii.astype(np.int64)
            out["in_ss_ids"] = np.array(sorted(iss), dtype=np.int64)
            for sid in iss:
                out["in_ss_tri_%d" % sid] = iss[sid].astype(np.int64)
        if c.get("golden_exo"):
            t, names, vals, conn, gcoord = read_golden_exo(
                os.path.join(d, c["golden_exo"]), inpoel.shape[0])
            # serial runs keep the input ordering of tets and nodes
            assert np.array_equal(conn, inpoel), name
            assert np.abs(gcoord - coord).max() < 1e-14, name
            # numerical fields (+ the per-element ndof of p-adaptive runs)
            keep = [i for i, n in enumerate(names) if n.endswith("_numerical") or n == "ndof"]
            out["exo_times"] = t
            out["exo_names"] = np.array([names[i] for i in keep])
            out["exo_vals"] = vals[:, keep, :]
            # every element field of the golden file, in the file's order (analytic and
            # err(.) fields of the manufactured-solution problems included)
            out["exo_names_all"] = np.array(names)
            out["exo_vals_all"] = vals
        for key, tag in (("golden_exo_chunks", "chunk"), ("golden_exo_chunks_overdecomposed", "ochunk")):
            if not c.get(key):
                continue
            # per-chare golden chunks of a partitioned run (partition-local order): keep, per tet,
            # its centroid and EVERY element field at the last output time; tests match tets to
            # the input mesh by centroid (Zoltan's assignment of tets to chares is not reproduced)
            cents, vals, tlast, names0 = [], [], None, None
            for fn in c[key]:
                f = netcdf_file(os.path.join(d, fn), "r", mmap=False)
                v = f.variables
                xyz = np.stack([v["coordx"][:], v["coordy"][:], v["coordz"][:]], axis=1).astype(np.float64)
                blk = [b for b in range(1, f.dimensions["num_el_blk"] + 1)
                       if v["connect%d" % b].elem_type.decode().upper().startswith("TET")][0]
                conn = np.array(v["connect%d" % blk][:], dtype=np.int64) - 1
                names = [_str(r) for r in v["name_elem_var"][:]]
                names0 = names0 or names
                assert names == names0
                cents.append(xyz[conn].mean(axis=1))
                vals.append(np.stack([np.array(v["vals_elem_var%deb%d" % (i + 1, blk)][:], dtype=np.float64)[-1]
                                      for i in range(len(names))]))
                tlast = float(v["time_whole"][:][-1])
                f.close()
            out[tag + "_centroid"] = np.concatenate(cents)
            out[tag + "_names"] = np.array(names0)
            out[tag + "_vals_last"] = np.concatenate(vals, axis=1)            # [nfield, ntet]
            out[tag + "_sizes"] = np.array([len(x) for x in cents])
            out[tag + "_time_last"] = np.array([tlast])
            if "c0_numerical" in names0:                                       # (kept for the transport test)
                out[tag + "_c0_last"] = out[tag + "_vals_last"][names0.index("c0_numerical")]
        if c.get("golden_exo_series"):
            # a run with mesh refinement during time stepping writes one file per mesh: keep, per
            # series, the refined mesh itself (coordinates + tets: this is what pins the 1:8 child
            # pattern), the output times and every element field
            for k, fn in enumerate(c["golden_exo_series"]):
                f = netcdf_file(os.path.join(d, fn), "r", mmap=False)
                v = f.variables
                blk = [b for b in range(1, f.dimensions["num_el_blk"] + 1)
                       if v["connect%d" % b].elem_type.decode().upper().startswith("TET")][0]
                names = [_str(r) for r in v["name_elem_var"][:]]
                out["s%d_coord" % k] = np.stack([v["coordx"][:], v["coordy"][:], v["coordz"][:]], axis=1).astype(np.float64)
                out["s%d_inpoel" % k] = np.array(v["connect%d" % blk][:], dtype=np.int64) - 1
                out["s%d_times" % k] = np.array(v["time_whole"][:], dtype=np.float64)
                out["s%d_vals" % k] = np.stack([np.array(v["vals_elem_var%deb%d" % (i + 1, blk)][:], dtype=np.float64)
                                               for i in range(len(names))], axis=1)      # [time, var, elem]
                out["series_names"] = np.array(names)
                f.close()
        if c.get("golden_diag"):
            out["diag"] = read_diag(os.path.join(d, c["golden_diag"]))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s ne=%6d nnode=%5d ss=%s -> %s (%.0f kB)" % (
            name, inpoel.shape[0], coord.shape[0], sorted(ss),
            os.path.basename(path), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    sys.exit(main())
