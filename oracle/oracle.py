"""ctypes front-end of the CPU oracle (oracle/dg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's `cpu_baseline` leg.  Nothing under quinoa_amd/ imports this module.

It assembles, with the oracle's own restatements, everything a serial
reference run holds for one mesh chunk (the `FaceData` arrays, geoFace,
geoElem, the regenerated boundary faces) and drives the reference's time loop
(limit -> dt -> rhs -> RK3 update).  Reference citations are in dg_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLUX = {"hllc": 0, "laxfriedrichs": 1}
LIMITER = {"nolimiter": 0, "wenop1": 1, "superbeep1": 2}
PROBLEM = {"user_defined": 0, "sod_shocktube": 1, "sedov_blastwave": 2,
           "vortical_flow": 3, "taylor_green": 4, "rotated_sod_shocktube": 6,
           "nl_energy_growth": 7, "rayleigh_taylor": 10}

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)


class Cfg(C.Structure):
    _fields_ = [("ndof", C.c_int64), ("rdof", C.c_int64), ("flux", C.c_int32),
                ("limiter", C.c_int32), ("problem", C.c_int32),
                ("pad_", C.c_int32), ("cweight", C.c_double),
                ("gamma", C.c_double), ("pstiff", C.c_double),
                ("cv", C.c_double), ("alpha", C.c_double),
                ("beta", C.c_double), ("p0", C.c_double),
                ("betax", C.c_double), ("betay", C.c_double), ("betaz", C.c_double),
                ("r0", C.c_double), ("ce", C.c_double), ("kappa", C.c_double)]


class Bc(C.Structure):
    _fields_ = [("nset", C.c_int64), ("set_id", c_i64p), ("set_off", c_i64p),
                ("set_face", c_i64p), ("ndir", C.c_int64), ("nsym", C.c_int64),
                ("nextrap", C.c_int64), ("dir", c_i64p), ("sym", c_i64p),
                ("extrap", c_i64p)]


def build(force=False):
    """Compile oracle/libdgoracle.so (and oracle/_ref when the reference is
    present, i.e. in the development container only)."""
    so = os.path.join(_HERE, "libdgoracle.so")
    src = os.path.join(_HERE, "dg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/src"):
        ref = os.path.join(_HERE, "_ref", "libquinoa_ref.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_dt.restype = C.c_double
        _LIB.orc_step.restype = C.c_double
        _LIB.orc_step_pdg.restype = C.c_double
        _LIB.orc_jacobian.restype = C.c_double
        _LIB.orc_gen_nipfac.restype = C.c_int64
        _LIB.orc_field_count.restype = C.c_int64
    return _LIB


def ref_lib():
    """The reference's own Vector.cpp/Quadrature.cpp build, or None."""
    p = os.path.join(_HERE, "_ref", "libquinoa_ref.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.ref_jacobian.restype = C.c_double
    return L


def _p(a, t):
    return a.ctypes.data_as(t)


def make_cfg(ndof, rdof=None, flux="hllc", limiter="nolimiter",
             problem="sod_shocktube", gamma=1.4, pstiff=0.0, cv=717.5,
             cweight=1.0, alpha=0.0, beta=0.0, p0=0.0, betax=0.0, betay=0.0, betaz=0.0,
             r0=0.0, ce=0.0, kappa=0.0):
    return Cfg(betax=betax, betay=betay, betaz=betaz, r0=r0, ce=ce, kappa=kappa,
               ndof=ndof, rdof=rdof or ndof, flux=FLUX[flux],
               limiter=LIMITER[limiter], problem=PROBLEM[problem], pad_=0,
               cweight=cweight, gamma=gamma, pstiff=pstiff, cv=cv,
               alpha=alpha, beta=beta, p0=p0)


def nleg_params(case):
    """nl_energy_growth parameters of a tests/golden case (0 when absent)"""
    return {k: case.get(k, 0.0) for k in ("betax", "betay", "betaz", "r0", "ce", "kappa")}


def regen_boundary_faces(inpoel, sidesets):
    """Boundary-face regeneration of the reference's mesh loader
    (src/Inciter/Partitioner.cpp:357-393): the file's side-set triangles are
    only order-independent keys; for each tet in input order its faces
    {0,2,1},{0,1,3},{0,3,2},{1,2,3} that match a side-set triangle are
    appended (in that node order) to that side set.  Boundary faces are then
    numbered side set by side set in ascending id (std::map<int,..> order of
    FaceData::m_bface, src/Inciter/FaceData.hpp).

    Returns (bface: {id: face ids}, triinpoel[nb,3])."""
    key = {}
    for sid in sorted(sidesets):
        for t in np.asarray(sidesets[sid]):
            key[tuple(sorted(int(v) for v in t))] = sid
    per = {sid: [] for sid in sorted(sidesets)}
    loc = ((0, 2, 1), (0, 1, 3), (0, 3, 2), (1, 2, 3))
    for t in np.asarray(inpoel):
        for f in loc:
            tri = (int(t[f[0]]), int(t[f[1]]), int(t[f[2]]))
            sid = key.get(tuple(sorted(tri)))
            if sid is not None:
                per[sid].append(tri)
    bface, tri_all, n = {}, [], 0
    for sid in sorted(per):
        m = len(per[sid])
        bface[sid] = np.arange(n, n + m, dtype=np.int64)
        tri_all.extend(per[sid])
        n += m
    triinpoel = np.array(tri_all, dtype=np.int64).reshape(-1, 3)
    return bface, triinpoel


class OracleMesh:
    """Serial (single-chunk) mesh with the reference's derived data."""

    def __init__(self, coord, inpoel, sidesets=None, bface=None, triinpoel=None):
        L = lib()
        self.coord = np.ascontiguousarray(coord, dtype=np.float64)
        self.x = np.ascontiguousarray(self.coord[:, 0])
        self.y = np.ascontiguousarray(self.coord[:, 1])
        self.z = np.ascontiguousarray(self.coord[:, 2])
        self.inpoel = np.ascontiguousarray(inpoel, dtype=np.int64).reshape(-1, 4)
        ne = self.nelem = self.inpoel.shape[0]
        npoin = self.npoin = self.coord.shape[0]
        if bface is None:
            bface, triinpoel = regen_boundary_faces(self.inpoel, sidesets or {})
        self.bface = {int(k): np.asarray(v, dtype=np.int64) for k, v in bface.items()}
        self.triinpoel = np.ascontiguousarray(triinpoel, dtype=np.int64).reshape(-1, 3)
        nb = self.nbfac = self.triinpoel.shape[0]
        inp = self.inpoel.reshape(-1)
        esup1 = np.zeros(4 * ne + 1, dtype=np.int64)
        esup2 = np.zeros(npoin + 1, dtype=np.int64)
        L.orc_gen_esup(_p(inp, c_i64p), C.c_int64(ne), C.c_int64(npoin),
                       _p(esup1, c_i64p), _p(esup2, c_i64p))
        self.esup1, self.esup2 = esup1, esup2
        self.esuel = np.zeros(4 * ne, dtype=np.int32)
        L.orc_gen_esuel(_p(inp, c_i64p), C.c_int64(ne), C.c_int64(npoin),
                        _p(esup1, c_i64p), _p(esup2, c_i64p), _p(self.esuel, c_i32p))
        nf = self.nfac = int(L.orc_gen_nipfac(C.c_int64(nb), _p(self.esuel, c_i32p), C.c_int64(ne)))
        self.inpofa = np.zeros(3 * nf, dtype=np.int64)
        tri = np.ascontiguousarray(self.triinpoel.reshape(-1))
        L.orc_gen_inpofa(C.c_int64(nb), _p(inp, c_i64p), C.c_int64(ne),
                         _p(tri, c_i64p), _p(self.esuel, c_i32p), _p(self.inpofa, c_i64p))
        self.belem = np.zeros(max(nb, 1), dtype=np.int64)
        L.orc_gen_belem(C.c_int64(nb), _p(self.inpofa, c_i64p), _p(esup1, c_i64p),
                        _p(esup2, c_i64p), _p(self.belem, c_i64p))
        self.belem = self.belem[:nb]
        self.esuf = np.zeros(2 * nf, dtype=np.int32)
        L.orc_gen_esuf(C.c_int64(nb), _p(self.belem, c_i64p), _p(self.esuel, c_i32p),
                       C.c_int64(ne), _p(self.esuf, c_i32p))
        self.geoFace = np.zeros(7 * nf)
        L.orc_gen_geoface(C.c_int64(nf), _p(self.inpofa, c_i64p), _p(self.x, c_f64p),
                          _p(self.y, c_f64p), _p(self.z, c_f64p), _p(self.geoFace, c_f64p))
        self.geoElem = np.zeros(4 * ne)
        L.orc_gen_geoelem(_p(inp, c_i64p), C.c_int64(ne), _p(self.x, c_f64p),
                          _p(self.y, c_f64p), _p(self.z, c_f64p), _p(self.geoElem, c_f64p))
        self.meshvol = float(self.geoElem[0::4].sum())
        # flattened bface map
        ids = sorted(self.bface)
        self._set_id = np.array(ids, dtype=np.int64)
        self._set_off = np.zeros(len(ids) + 1, dtype=np.int64)
        faces = []
        for i, s in enumerate(ids):
            faces.append(self.bface[s])
            self._set_off[i + 1] = self._set_off[i] + len(self.bface[s])
        self._set_face = (np.concatenate(faces) if faces else np.zeros(0, np.int64)).astype(np.int64)
        if self._set_face.size == 0:
            self._set_face = np.zeros(1, dtype=np.int64)


class ChunkMesh:
    """Oracle view of a chunk WITH ghosts, built from arrays in the reference's
    data model (inpoel[4*nunk], esuel[4*nielem], esuf/inpofa/geoFace incl. the
    chare-boundary faces, geoElem[4*nunk], bface).  Used by the multi-rank
    tests: the reference's ghost set-up (src/Inciter/DG.cpp:134-949) is not
    restated by the oracle, so these arrays come from the chunk under test."""

    def __init__(self, coord, inpoel, nielem, esuel, esuf, inpofa, geoFace, geoElem, bface, nbfac):
        self.coord = np.ascontiguousarray(coord, dtype=np.float64)
        self.x = np.ascontiguousarray(self.coord[:, 0])
        self.y = np.ascontiguousarray(self.coord[:, 1])
        self.z = np.ascontiguousarray(self.coord[:, 2])
        self.inpoel = np.ascontiguousarray(inpoel, dtype=np.int64).reshape(-1, 4)
        self.nelem = self.inpoel.shape[0]          # nunk
        self.nielem = int(nielem)
        self.npoin = self.coord.shape[0]
        self.esuel = np.ascontiguousarray(esuel, dtype=np.int32)
        self.esuf = np.ascontiguousarray(esuf, dtype=np.int32)
        self.inpofa = np.ascontiguousarray(inpofa, dtype=np.int64)
        self.geoFace = np.ascontiguousarray(geoFace, dtype=np.float64)
        self.geoElem = np.ascontiguousarray(geoElem, dtype=np.float64)
        self.nbfac, self.nfac = int(nbfac), len(self.esuf) // 2
        self.bface = {int(k): np.asarray(v, dtype=np.int64) for k, v in bface.items()}
        self.meshvol = float(self.geoElem[0:4 * self.nielem:4].sum())
        ids = sorted(self.bface)
        self._set_id = np.array(ids or [0], dtype=np.int64)[:len(ids)] if ids else np.zeros(0, np.int64)
        self._set_off = np.zeros(len(ids) + 1, dtype=np.int64)
        faces = []
        for i, s in enumerate(ids):
            faces.append(self.bface[s])
            self._set_off[i + 1] = self._set_off[i] + len(self.bface[s])
        self._set_face = (np.concatenate(faces) if faces else np.zeros(1, np.int64)).astype(np.int64)
        if self._set_face.size == 0:
            self._set_face = np.zeros(1, dtype=np.int64)
        if self._set_id.size == 0:
            self._set_id = np.zeros(1, dtype=np.int64)
            self._nset = 0
        else:
            self._nset = len(ids)


class Oracle:
    """Reference time loop on one mesh chunk (CPU, AoS fields)."""

    def __init__(self, mesh, cfg, bc_dirichlet=(), bc_sym=(), bc_extrapolate=(), pref=False,
                 tolref=0.1):
        self.m, self.cfg, self.L_ = mesh, cfg, lib()
        # p-adaptive DG (scheme pdg): DG::m_ndof, all elements start at ndof (DG.cpp:927)
        self.pref, self.tolref = bool(pref), float(tolref)
        self.ndofel = np.full(mesh.nelem, cfg.ndof, dtype=np.int64) if pref else None
        self._dir = np.array(list(bc_dirichlet) or [0], dtype=np.int64)
        self._sym = np.array(list(bc_sym) or [0], dtype=np.int64)
        self._ext = np.array(list(bc_extrapolate) or [0], dtype=np.int64)
        self.nie = getattr(mesh, "nielem", mesh.nelem)
        self.bc = Bc(nset=getattr(mesh, "_nset", len(mesh._set_id)), set_id=_p(mesh._set_id, c_i64p),
                     set_off=_p(mesh._set_off, c_i64p), set_face=_p(mesh._set_face, c_i64p),
                     ndir=len(bc_dirichlet), nsym=len(bc_sym), nextrap=len(bc_extrapolate),
                     dir=_p(self._dir, c_i64p), sym=_p(self._sym, c_i64p),
                     extrap=_p(self._ext, c_i64p))
        self.nprop = 5 * cfg.rdof
        self.npropr = 5 * cfg.ndof

    class _Ndofel:
        """operators see the per-element ndof while inside (orc_set_ndofel)"""
        def __init__(self, o):
            self.o = o
        def __enter__(self):
            if self.o.pref:
                self.o.L_.orc_set_ndofel(_p(self.o.ndofel, c_i64p))
        def __exit__(self, *a):
            if self.o.pref:
                self.o.L_.orc_set_ndofel(None)

    # --- single operators -------------------------------------------------
    def lhs(self):
        m = self.m
        Lm = np.zeros(m.nelem * self.npropr)
        self.L_.orc_mass(C.byref(self.cfg), _p(m.geoElem, c_f64p), C.c_int64(m.nelem), _p(Lm, c_f64p))
        return Lm

    def initialize(self, Lm, t=0.0):
        m = self.m
        U = np.zeros(m.nelem * self.nprop)
        self.L_.orc_initialize(C.byref(self.cfg), _p(Lm, c_f64p), _p(m.inpoel.reshape(-1), c_i64p),
                               _p(m.x, c_f64p), _p(m.y, c_f64p), _p(m.z, c_f64p),
                               _p(U, c_f64p), C.c_double(t), C.c_int64(self.nie))
        return U

    def rhs(self, t, U):
        m = self.m
        R = np.zeros(m.nelem * self.npropr)
        with self._Ndofel(self):
          self.L_.orc_rhs(C.byref(self.cfg), C.byref(self.bc), C.c_double(t), C.c_int64(m.nelem),
                          C.c_int64(m.nbfac), C.c_int64(m.nfac), _p(m.esuf, c_i32p),
                          _p(m.inpofa, c_i64p), _p(m.inpoel.reshape(-1), c_i64p),
                          _p(m.x, c_f64p), _p(m.y, c_f64p), _p(m.z, c_f64p),
                          _p(m.geoFace, c_f64p), _p(m.geoElem, c_f64p), _p(U, c_f64p), _p(R, c_f64p))
        return R

    def dt(self, U):
        m = self.m
        with self._Ndofel(self):
          return float(self.L_.orc_dt(C.byref(self.cfg), C.c_int64(m.nelem), C.c_int64(m.nfac),
                                    _p(m.esuf, c_i32p), _p(m.inpofa, c_i64p),
                                    _p(m.inpoel.reshape(-1), c_i64p), _p(m.x, c_f64p),
                                    _p(m.y, c_f64p), _p(m.z, c_f64p), _p(m.geoFace, c_f64p),
                                    _p(m.geoElem, c_f64p), _p(U, c_f64p)))

    def limit(self, U, esuel=None, nrows=None):
        """In place, like the reference (src/Inciter/DG.cpp:1251-1260).  esuel / nrows: the same limiter over the
        first nrows rows with a caller-supplied esuel[4 * nrows] -- the multi-rank tests limit the layer-1 ghosts
        of a chunk with two ghost layers this way (the limiters read esuel and the solution only,
        src/PDE/Limiter.cpp:29-316)."""
        m = self.m
        es = m.esuel if esuel is None else np.ascontiguousarray(esuel, dtype=np.int32).reshape(-1)
        nr = self.nie if nrows is None else int(nrows)
        with self._Ndofel(self):
            self.L_.orc_limit(C.byref(self.cfg), _p(es, c_i32p), C.c_int64(nr),
                              _p(m.inpoel.reshape(-1), c_i64p), _p(m.x, c_f64p), _p(m.y, c_f64p),
                              _p(m.z, c_f64p), _p(U, c_f64p))
        return U

    def rk_update(self, stage, dt, Un, R, Lm, U):
        self.L_.orc_rk_update(C.byref(self.cfg), C.c_int(stage), C.c_double(dt), _p(Un, c_f64p),
                              _p(R, c_f64p), _p(Lm, c_f64p), _p(U, c_f64p), C.c_int64(self.m.nelem))
        return U

    def diag(self, t_new, U):
        """Row of the reference's diag file after sqrt(./V): L2(u_c) x5,
        L2(u_c - analytic) x5 (Transporter.cpp:901-916)."""
        m = self.m
        out = np.zeros(15)
        with self._Ndofel(self):
            self.L_.orc_diag(C.byref(self.cfg), C.c_double(t_new), _p(m.inpoel.reshape(-1), c_i64p),
                             _p(m.x, c_f64p), _p(m.y, c_f64p), _p(m.z, c_f64p),
                             _p(m.geoElem, c_f64p), _p(U, c_f64p), C.c_int64(self.nie), _p(out, c_f64p))
        return np.sqrt(out[:10] / m.meshvol), out[10:]

    def step(self, t, U, Lm, fixed_dt=0.0, cfl=0.0, tleft=1e300, work=None):
        m = self.m
        if work is None:
            work = (np.zeros_like(U), np.zeros(m.nelem * self.npropr))
        Un, R = work
        if self.pref:
            dt = self.L_.orc_step_pdg(
                C.byref(self.cfg), C.byref(self.bc), C.c_double(t), C.c_double(fixed_dt),
                C.c_double(cfl), C.c_double(tleft), C.c_double(self.tolref), C.c_int64(m.nelem),
                C.c_int64(m.nbfac), C.c_int64(m.nfac), _p(m.esuel, c_i32p), _p(m.esuf, c_i32p),
                _p(m.inpofa, c_i64p), _p(m.inpoel.reshape(-1), c_i64p), _p(m.x, c_f64p),
                _p(m.y, c_f64p), _p(m.z, c_f64p), _p(m.geoFace, c_f64p), _p(m.geoElem, c_f64p),
                _p(Lm, c_f64p), _p(U, c_f64p), _p(Un, c_f64p), _p(R, c_f64p), _p(self.ndofel, c_i64p))
            return float(dt)
        dt = self.L_.orc_step(C.byref(self.cfg), C.byref(self.bc), C.c_double(t),
                              C.c_double(fixed_dt), C.c_double(cfl), C.c_double(tleft),
                              C.c_int64(m.nelem), C.c_int64(m.nbfac), C.c_int64(m.nfac),
                              _p(m.esuel, c_i32p), _p(m.esuf, c_i32p), _p(m.inpofa, c_i64p),
                              _p(m.inpoel.reshape(-1), c_i64p), _p(m.x, c_f64p), _p(m.y, c_f64p),
                              _p(m.z, c_f64p), _p(m.geoFace, c_f64p), _p(m.geoElem, c_f64p),
                              _p(Lm, c_f64p), _p(U, c_f64p), _p(Un, c_f64p), _p(R, c_f64p))
        return float(dt)

    # --- p-adaptive DG pieces, for runs that exchange ghosts between them ---
    def eval_ndof(self, U, tolref=None):
        """DG::eval_ndof over the owned tets (DG.cpp:1088-1163)"""
        m = self.m
        self.L_.orc_eval_ndof(C.byref(self.cfg), C.c_int64(self.nie), _p(m.inpoel.reshape(-1), c_i64p),
                              _p(m.x, c_f64p), _p(m.y, c_f64p), _p(m.z, c_f64p), _p(U, c_f64p),
                              C.c_double(self.tolref if tolref is None else tolref), _p(self.ndofel, c_i64p))

    def propagate_ndof(self):
        """DG::propagate_ndof across interior and chare-boundary faces (DG.cpp:1284-1313)"""
        m = self.m
        self.L_.orc_propagate_ndof(C.c_int64(m.nelem), C.c_int64(m.nbfac), C.c_int64(m.nfac),
                                   _p(m.esuf, c_i32p), _p(self.ndofel, c_i64p))

    def pdg_zero(self, U):
        """DG::solve, stage 0: zero the high-order DOFs of P0 tets (DG.cpp:1451-1469)"""
        self.L_.orc_pdg_zero(C.byref(self.cfg), C.c_int64(self.m.nelem), _p(self.ndofel, c_i64p), _p(U, c_f64p))

    # --- field output (cell means), Problem::fieldOutput ------------------
    def field_output(self, U):
        """density, x/y/z velocity, specific total energy, pressure from cell
        means (src/PDE/CompFlow/Problem/SodShocktube.cpp:160-215)."""
        rd = self.cfg.rdof
        Um = U.reshape(self.m.nelem, self.nprop)
        r, ru, rv, rw, re = (Um[:, c * rd] for c in range(5))
        u, v, w = ru / r, rv / r, rw / r
        g, pc = self.cfg.gamma, self.cfg.pstiff
        p = (re - 0.5 * r * (u * u + v * v + w * w) - pc) * (g - 1.0) - pc
        return np.stack([r, u, v, w, re / r, p])

    FIELD_NAMES = {
        3: ["density_numerical", "density_analytical", "x-velocity_numerical", "x-velocity_analytical",
            "y-velocity_numerical", "y-velocity_analytical", "z-velocity_numerical",
            "z-velocity_analytical", "specific_total_energy_numerical",
            "specific_total_energy_analytical", "pressure_numerical", "pressure_analytical"],
        4: ["density_numerical", "density_analytical", "x-velocity_numerical", "x-velocity_analytical",
            "err(u)", "y-velocity_numerical", "y-velocity_analytical", "err(v)", "z-velocity_numerical",
            "z-velocity_analytical", "specific_total_energy_numerical",
            "specific_total_energy_analytical", "err(E)", "pressure_numerical", "pressure_analytical"],
        7: ["density_numerical", "x-velocity_numerical", "y-velocity_numerical", "z-velocity_numerical",
            "specific_total_energy_numerical", "pressure_numerical", "density_analytical",
            "x-velocity_analytical", "y-velocity_analytical", "z-velocity_analytical",
            "specific_total_energy_analytical", "pressure_analytical", "err(rho)", "err(e)"],
        0: ["density", "x-velocity", "y-velocity", "z-velocity", "specific total energy", "pressure",
            "temperature"],
    }
    FIELD_NAMES[10] = FIELD_NAMES[7] + ["err(p)", "err(u)", "err(v)", "err(w)"]

    def field_names(self):
        """Problem::fieldNames (src/PDE/CompFlow/Problem/*.cpp)"""
        six = ["density_numerical", "x-velocity_numerical", "y-velocity_numerical",
               "z-velocity_numerical", "specific_total_energy_numerical", "pressure_numerical"]
        return list(self.FIELD_NAMES.get(int(self.cfg.problem), six))

    def field_output_all(self, U, t):
        """Problem::fieldOutput as dg::CompFlow::fieldOutput calls it (V = 0): every field of
        the Problem's list, [nfield, nelem]"""
        nf = int(self.L_.orc_field_count(C.byref(self.cfg)))
        out = np.zeros((nf, self.m.nelem))
        Uc = np.ascontiguousarray(U, dtype=np.float64)
        with np.errstate(all="ignore"):
            self.L_.orc_field_output(C.byref(self.cfg), C.c_double(t), C.c_int64(self.m.nelem),
                                     _p(self.m.geoElem, c_f64p), _p(Uc, c_f64p), _p(out, c_f64p))
        return out

    def avg_elem_to_node(self, U):
        """dg::CompFlow::avgElemToNode (DGCompFlow.hpp:465-552): [6, npoin]"""
        out = np.zeros((6, self.m.npoin))
        Uc = np.ascontiguousarray(U, dtype=np.float64)
        inp = np.ascontiguousarray(self.m.inpoel.reshape(-1))
        self.L_.orc_avg_elem_to_node(C.byref(self.cfg), _p(inp, c_i64p), C.c_int64(self.m.nelem),
                                     C.c_int64(self.m.npoin), _p(self.m.x, c_f64p), _p(self.m.y, c_f64p),
                                     _p(self.m.z, c_f64p), _p(Uc, c_f64p), _p(out, c_f64p))
        return out


def run_case(case, fix, nstep=None, on_step=None):
    """Run one tests/golden case with the oracle.  Returns dict with the diag
    rows and the field output at every plot time (incl. t=0 and last step)."""
    mesh = OracleMesh(fix["coord"], fix["inpoel"],
                      {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]})
    cfg = make_cfg(case["ndof"], flux=case["flux"], limiter=case["limiter"],
                   problem=case["problem"], gamma=case["gamma"],
                   alpha=case.get("alpha", 0.0), beta=case.get("beta", 0.0),
                   p0=case.get("p0", 0.0), **nleg_params(case))
    orc = Oracle(mesh, cfg, case["bc_dirichlet"], case["bc_sym"], case["bc_extrapolate"],
                 pref=case.get("pref", False), tolref=case.get("tolref", 0.1))
    Lm = orc.lhs()
    U = orc.initialize(Lm, 0.0)
    t, it = 0.0, 0
    nstep = nstep or case["nstep"]
    diag_rows, fields, times = [], [orc.field_output(U)], [0.0]
    fields_all = [orc.field_output_all(U, 0.0)]
    ndofs = [orc.ndofel.copy()] if orc.pref else []
    work = (np.zeros_like(U), np.zeros(mesh.nelem * orc.npropr))
    while it < nstep:
        dt = orc.step(t, U, Lm, fixed_dt=case["dt"], cfl=case["cfl"], work=work)
        if (it + 1) % case["diag_interval"] == 0:
            l2, _ = orc.diag(t + dt, U)
            diag_rows.append(np.concatenate([[it + 1, t + dt, dt], l2]))
        t += dt
        it += 1
        if it % case["plot_interval"] == 0 or it == nstep:
            fields.append(orc.field_output(U))
            fields_all.append(orc.field_output_all(U, t))
            times.append(t)
            if orc.pref:
                ndofs.append(orc.ndofel.copy())
        if on_step:
            on_step(it, t, dt, U)
    return {"mesh": mesh, "oracle": orc, "U": U, "L": Lm, "t": t,
            "diag": np.array(diag_rows), "fields": np.array(fields), "fields_all": np.array(fields_all),
            "times": np.array(times), "ndof": np.array(ndofs)}


# ---------------------------------------------------------------- transport
TR_PROBLEM = {"slot_cyl": 1, "cyl_advect": 2, "gauss_hump": 3, "shear_diff": 4}


def tr_select_component(case, c):
    """Select scalar c of case["ncomp"] for the orc_tr_* functions (the scalars of a dg::Transport
    system never couple: DGTransport.hpp:129-186 loops over them inside every integrator), with
    its shear_diff parameters u0[c], lambda[2c:2c+2], diffusivity[3c:3c+3] (ShearDiff.cpp:43-68)."""
    L = lib()
    L.orc_tr_set_component(C.c_int(c), C.c_int(int(case.get("ncomp", 1))))
    if case.get("problem") == "shear_diff":
        lam = np.ascontiguousarray(case["lambda"][2 * c:2 * c + 2], dtype=np.float64)
        dif = np.ascontiguousarray(case["diffusivity"][3 * c:3 * c + 3], dtype=np.float64)
        L.orc_tr_set_shear_diff(C.c_double(float(case["u0"][c])), _p(lam, c_f64p), _p(dif, c_f64p))


def run_transport_multi(case, fix, nstep=None):
    """dg::Transport with case["ncomp"] scalars: one run_transport_case per scalar (see
    tr_select_component), rows assembled component-major, U[e, c*ndof + k] (mark = c*rdof).
    case["t0"]: start time (shear_diff's solution is singular at t = 0)."""
    nc, ndof = int(case.get("ncomp", 1)), case["ndof"]
    runs = []
    try:
        for c in range(nc):
            tr_select_component(case, c)
            runs.append(run_transport_case(case, fix, nstep=nstep, t0=float(case.get("t0", 0.0))))
    finally:
        lib().orc_tr_set_component(C.c_int(0), C.c_int(1))
    ne = runs[0]["mesh"].nelem
    U = np.zeros((ne, nc * ndof))
    for c, r in enumerate(runs):
        U[:, c * ndof:(c + 1) * ndof] = r["U"].reshape(ne, ndof)
    return {"mesh": runs[0]["mesh"], "U": U.reshape(-1), "t": runs[0]["t"], "runs": runs,
            "diag": [r["diag"] for r in runs]}


def run_transport_case(case, fix, nstep=None, U0=None, t0=0.0, it0=0):
    """dg::Transport with one scalar (src/PDE/Transport/DGTransport.hpp:129-186,
    Upwind flux) on one chunk, fixed dt, in the DG chare's stage order: limiter
    (WENO_P1 / Superbee_P1 with ncomp = 1), rhs, SSP-RK3; with scheme pdg also
    eval_ndof / propagate_ndof / zeroing at stage 0.  BASELINE config 1 is the
    slot_cyl DG-P0 case.  U0 / t0 / it0: continue from a given state (after a mesh refinement:
    the state DG::resizePostAMR hands over) instead of projecting the initial condition."""
    L = lib()
    m = OracleMesh(fix["coord"], fix["inpoel"],
                   {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]})
    ndof = case["ndof"]
    ne = m.nelem
    prob = C.c_int(TR_PROBLEM[case.get("problem", "slot_cyl")])
    limiter = case.get("limiter", "nolimiter")
    pref, tolref = bool(case.get("pref", False)), float(case.get("tolref", 0.1))
    # BC type per side set in the bface map (DGTransport.hpp:163-168 order)
    types = {"bc_extrapolate": 0, "bc_inlet": 1, "bc_outlet": 2, "bc_dirichlet": 3}
    bctype = np.full(len(m._set_id), -1, dtype=np.int32)
    for key, t in types.items():
        for sid in case.get(key, []):
            bctype[list(m._set_id).index(sid)] = t
    zero = np.zeros(1, dtype=np.int64)
    bc = Bc(nset=len(m._set_id), set_id=_p(m._set_id, c_i64p), set_off=_p(m._set_off, c_i64p),
            set_face=_p(m._set_face, c_i64p), ndir=0, nsym=0, nextrap=0,
            dir=_p(zero, c_i64p), sym=_p(zero, c_i64p), extrap=_p(zero, c_i64p))
    inp = m.inpoel.reshape(-1)
    mesh_args = (_p(inp, c_i64p), _p(m.x, c_f64p), _p(m.y, c_f64p), _p(m.z, c_f64p))
    Lm = np.zeros(ne * ndof)
    L.orc_tr_mass(C.c_int64(ndof), _p(m.geoElem, c_f64p), C.c_int64(ne), _p(Lm, c_f64p))
    U = np.zeros(ne * ndof)
    L.orc_tr_initialize(prob, C.c_int64(ndof), _p(Lm, c_f64p), _p(inp, c_i64p), _p(m.x, c_f64p),
                        _p(m.y, c_f64p), _p(m.z, c_f64p), _p(U, c_f64p),
                        C.c_double(0.0 if U0 is not None else float(t0)), C.c_int64(ne))
    if U0 is not None:
        U[:] = np.asarray(U0, dtype=np.float64).reshape(-1)
    ndofel = np.full(ne, ndof, dtype=np.int64)
    Un, R = np.zeros_like(U), np.zeros_like(U)
    t, dt, rows = float(t0), case["dt"], []
    nstep = case["nstep"] if nstep is None else nstep
    plot = case.get("plot_interval", 1)
    means = lambda: U.reshape(ne, ndof)[:, 0].copy()
    fields, times, ndofs = [means()], [t], [ndofel.copy()]
    if pref:
        L.orc_set_ndofel(_p(ndofel, c_i64p))
    try:
        for it in range(it0, it0 + nstep):
            for stage in range(3):
                if pref and stage == 0:
                    L.orc_tr_eval_ndof(C.c_int64(ndof), C.c_int64(ne), *mesh_args, _p(U, c_f64p),
                                       C.c_double(tolref), _p(ndofel, c_i64p))
                    L.orc_propagate_ndof(C.c_int64(ne), C.c_int64(m.nbfac), C.c_int64(m.nfac),
                                         _p(m.esuf, c_i32p), _p(ndofel, c_i64p))
                if limiter == "superbeep1":
                    L.orc_tr_superbee(C.c_int64(ndof), _p(m.esuel, c_i32p), C.c_int64(ne), *mesh_args,
                                      _p(U, c_f64p))
                elif limiter == "wenop1":
                    L.orc_tr_weno(C.c_int64(ndof), C.c_double(case.get("cweight", 1.0)),
                                  _p(m.esuel, c_i32p), C.c_int64(ne), _p(U, c_f64p))
                if stage == 0:
                    if pref:
                        L.orc_tr_pdg_zero(C.c_int64(ndof), C.c_int64(ne), _p(ndofel, c_i64p), _p(U, c_f64p))
                    Un[:] = U
                L.orc_tr_rhs(prob, C.c_int64(ndof), C.byref(bc), _p(bctype, c_i32p), C.c_double(t),
                             C.c_int64(ne), C.c_int64(m.nbfac), C.c_int64(m.nfac), _p(m.esuf, c_i32p),
                             _p(m.inpofa, c_i64p), *mesh_args, _p(m.geoFace, c_f64p),
                             _p(m.geoElem, c_f64p), _p(U, c_f64p), _p(R, c_f64p))
                L.orc_tr_rk_update(C.c_int64(ndof), C.c_int(stage), C.c_double(dt), _p(Un, c_f64p),
                                   _p(R, c_f64p), _p(Lm, c_f64p), _p(U, c_f64p), C.c_int64(ne))
            t += dt
            if (it + 1) % case.get("diag_interval", 1) == 0:
                out = np.zeros(3)
                L.orc_tr_diag(prob, C.c_int64(ndof), C.c_double(t), *mesh_args, _p(m.geoElem, c_f64p),
                              _p(U, c_f64p), C.c_int64(ne), _p(out, c_f64p))
                rows.append([it + 1, t, dt, np.sqrt(out[0] / m.meshvol), np.sqrt(out[1] / m.meshvol), out[2]])
            if (it + 1) % plot == 0 or it + 1 == it0 + nstep:
                fields.append(means()); times.append(t); ndofs.append(ndofel.copy())
    finally:
        if pref:
            L.orc_set_ndofel(None)
    return {"mesh": m, "U": U, "L": Lm, "diag": np.array(rows), "t": t, "fields": np.array(fields),
            "times": np.array(times), "ndof": np.array(ndofs)}
