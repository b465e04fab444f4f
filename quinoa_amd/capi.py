"""ctypes binding of the C ABI in include/qdg.h (quinoa_amd/lib/libqdg.so).

This is plumbing for tests and bench.py: the product is the shared library.
There is no fallback of any kind here -- if libqdg.so is missing or a call
fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# QDG_LIB selects an alternative build of the same library (kernel A/B runs)
LIB_PATH = os.environ.get("QDG_LIB") or os.path.join(_HERE, "lib", "libqdg.so")

FLUX = {"hllc": 0, "laxfriedrichs": 1, "upwind": 2}
LIMITER = {"nolimiter": 0, "wenop1": 1, "superbeep1": 2}
PROBLEM = {"user_defined": 0, "sod_shocktube": 1, "sedov_blastwave": 2,
           "vortical_flow": 3, "taylor_green": 4, "slot_cyl": 5,
           "rotated_sod_shocktube": 6, "nl_energy_growth": 7, "cyl_advect": 8, "gauss_hump": 9,
           "rayleigh_taylor": 10, "shear_diff": 11}
BC_DIRICHLET, BC_SYMMETRY, BC_EXTRAPOLATE, BC_INLET, BC_OUTLET = 1, 2, 3, 4, 5
PDE = {"compflow": 0, "transport": 1}

c_szp = C.POINTER(C.c_size_t)
c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)


class QdgError(RuntimeError):
    pass


class qdg_config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32),
                ("ndof", C.c_int32), ("rdof", C.c_int32), ("flux", C.c_int32),
                ("limiter", C.c_int32), ("problem", C.c_int32), ("nbc", C.c_int32),
                ("bc_sideset", c_i32p), ("bc_type", c_i32p),
                ("gamma", C.c_double), ("pstiff", C.c_double), ("cv", C.c_double),
                ("cweight", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("p0", C.c_double), ("cfl", C.c_double), ("dt", C.c_double),
                ("pde", C.c_int32), ("pref", C.c_int32), ("tolref", C.c_double),
                ("betax", C.c_double), ("betay", C.c_double), ("betaz", C.c_double),
                ("r0", C.c_double), ("ce", C.c_double), ("kappa", C.c_double),
                ("ncomp", C.c_int32), ("reserved0", C.c_int32),
                ("tr_u0", c_f64p), ("tr_lambda", c_f64p), ("tr_diffusivity", c_f64p)]


class qdg_bface(C.Structure):
    _fields_ = [("nset", C.c_size_t), ("set_id", c_i32p), ("set_off", c_szp),
                ("face", c_szp)]


_lib = None


def lib():
    """Load libqdg.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QdgError("HIP extension missing: %s (run `python -c 'import "
                           "__graft_entry__ as g; g.build()'`); there is no CPU "
                           "fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.qdg_last_error.restype = C.c_char_p
        L.qdg_version.restype = C.c_char_p
        L.qdg_gen_nipfac.restype = C.c_size_t
        L.qdg_gen_nipfac.argtypes = [C.c_size_t, C.c_size_t, c_i32p]
        _lib = L
    return _lib


def device_pool_trim():
    """qdg_device_pool_trim: hand the library's cache of freed device buffers back to the driver;
    returns the bytes released"""
    n = C.c_size_t()
    _chk(lib().qdg_device_pool_trim(C.byref(n)))
    return n.value


def _chk(rc):
    if rc != 0:
        raise QdgError(lib().qdg_last_error().decode())


def _sz(a):
    a = np.asarray(a)
    if a.dtype == np.int64 and a.flags.c_contiguous:
        a = a.view(np.uint64)              # ids are non-negative: same bits, no copy
    else:
        a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(c_szp)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_f64p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_i32p)


# ------------------------------------------------------------------ host mesh
# mirrors of inciter::FaceData / DerivedData (no GPU needed)

def bnd_faces(inpoel, sidesets):
    """Boundary-face regeneration (Partitioner.cpp:357-393).  `sidesets` maps
    side set id -> triangles[n,3].  Returns (bface {id: face ids}, triinpoel)."""
    L = lib()
    inp, pinp = _sz(np.asarray(inpoel).reshape(-1))
    ids = sorted(sidesets)
    tri = np.concatenate([np.asarray(sidesets[s]).reshape(-1, 3) for s in ids]) if ids \
        else np.zeros((0, 3), np.uint64)
    tset = np.concatenate([np.full(len(np.asarray(sidesets[s]).reshape(-1, 3)), s, np.int32)
                           for s in ids]) if ids else np.zeros(0, np.int32)
    ntri = tri.shape[0]
    tri, ptri = _sz(tri.reshape(-1))
    tset, ptset = _i32(tset)
    out_tri = np.zeros(3 * max(ntri, 1), dtype=np.uint64)
    out_set = np.zeros(max(ntri, 1), dtype=np.int32)
    nb = C.c_size_t(0)
    _chk(L.qdg_bnd_faces(C.c_size_t(len(inp) // 4), pinp, C.c_size_t(ntri), ptri, ptset,
                         C.byref(nb), out_tri.ctypes.data_as(c_szp),
                         out_set.ctypes.data_as(c_i32p)))
    nb = nb.value
    out_set = out_set[:nb]
    bface = {int(s): np.nonzero(out_set == s)[0].astype(np.uint64) for s in ids}
    return bface, out_tri[:3 * nb].reshape(-1, 3).copy()


class FaceData:
    """inciter::FaceData (src/Inciter/FaceData.hpp:41-106): same ctor arguments,
    same members (esuel, nipfac, inpofa, belem, esuf), built by libqdg."""

    def __init__(self, inpoel, bface, triinpoel):
        L = lib()
        self.inpoel, pinp = _sz(np.asarray(inpoel).reshape(-1))
        ne = self.nelem = len(self.inpoel) // 4
        self.bface = {int(k): np.ascontiguousarray(v, dtype=np.uint64) for k, v in bface.items()}
        self.triinpoel, ptri = _sz(np.asarray(triinpoel).reshape(-1))
        nb = self.nbfac = sum(len(v) for v in self.bface.values())
        assert nb == len(self.triinpoel) // 3
        self.esuel = np.zeros(4 * ne, dtype=np.int32)
        _chk(L.qdg_gen_esuel(C.c_size_t(ne), pinp, self.esuel.ctypes.data_as(c_i32p)))
        self.nipfac = int(L.qdg_gen_nipfac(C.c_size_t(ne), C.c_size_t(nb),
                                           self.esuel.ctypes.data_as(c_i32p)))
        self.inpofa = np.zeros(3 * self.nipfac, dtype=np.uint64)
        _chk(L.qdg_gen_inpofa(C.c_size_t(ne), C.c_size_t(nb), pinp, ptri,
                              self.esuel.ctypes.data_as(c_i32p), self.inpofa.ctypes.data_as(c_szp)))
        self.belem = np.zeros(max(nb, 1), dtype=np.uint64)
        _chk(L.qdg_gen_belem(C.c_size_t(ne), C.c_size_t(nb), pinp,
                             self.inpofa.ctypes.data_as(c_szp), self.belem.ctypes.data_as(c_szp)))
        self.esuf = np.zeros(2 * self.nipfac, dtype=np.int32)
        _chk(L.qdg_gen_esuf(C.c_size_t(ne), C.c_size_t(nb), self.belem.ctypes.data_as(c_szp),
                            self.esuel.ctypes.data_as(c_i32p), self.esuf.ctypes.data_as(c_i32p)))
        self.belem = self.belem[:nb]


def gen_esuel(inpoel):
    """qdg_gen_esuel: esuel[nelem, 4] (tet across local face lpofa[f], -1: none) of any tet connectivity"""
    inp, pinp = _sz(np.asarray(inpoel).reshape(-1))
    es = np.zeros(len(inp), dtype=np.int32)
    _chk(lib().qdg_gen_esuel(C.c_size_t(len(inp) // 4), pinp, es.ctypes.data_as(c_i32p)))
    return es.reshape(-1, 4)


def gen_geoface(nfac, inpofa, coord):
    x, px = _f64(coord[:, 0]); y, py = _f64(coord[:, 1]); z, pz = _f64(coord[:, 2])
    inpofa, pf = _sz(inpofa)
    g = np.zeros(7 * nfac)
    _chk(lib().qdg_gen_geoface(C.c_size_t(nfac), pf, px, py, pz, g.ctypes.data_as(c_f64p)))
    return g


def gen_geoelem(inpoel, coord):
    x, px = _f64(coord[:, 0]); y, py = _f64(coord[:, 1]); z, pz = _f64(coord[:, 2])
    inp, pinp = _sz(np.asarray(inpoel).reshape(-1))
    g = np.zeros(len(inp))
    _chk(lib().qdg_gen_geoelem(C.c_size_t(len(inp) // 4), pinp, px, py, pz,
                               g.ctypes.data_as(c_f64p)))
    return g


# ------------------------------------------------------------------ device side

# options applied to every new Context (qdg_ctx_set_option); tests use it to run a whole
# golden case through another kernel form
default_options = {}


class Context:
    def __init__(self, ndof, flux="hllc", limiter="nolimiter", problem="sod_shocktube",
                 gamma=1.4, pstiff=0.0, cv=717.5, cweight=1.0, alpha=0.0, beta=0.0, p0=0.0,
                 cfl=0.0, dt=0.0, bc_dirichlet=(), bc_sym=(), bc_extrapolate=(), device=0,
                 pde="compflow", bc_inlet=(), bc_outlet=(), pref=False, tolref=0.1,
                 betax=0.0, betay=0.0, betaz=0.0, r0=0.0, ce=0.0, kappa=0.0, options=None,
                 ncomp=None, u0=None, lam=None, diffusivity=None):
        """ncomp: transported scalars of a dg::Transport system (default 1); u0 [ncomp], lam
        [2*ncomp], diffusivity [3*ncomp]: the shear_diff parameters (param::transport::u0 |
        lambda | diffusivity)."""
        L = lib()
        nc = int(ncomp) if ncomp else (1 if pde == "transport" else 5)
        self._sd = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None
                    for a in (u0, lam, diffusivity)]
        for a, n in zip(self._sd, (nc, 2 * nc, 3 * nc)):
            if a is not None and a.size != n:
                raise QdgError("shear_diff parameters: expected %d values, got %d" % (n, a.size))
        psd = [a.ctypes.data_as(c_f64p) if a is not None else None for a in self._sd]
        ss = list(bc_dirichlet) + list(bc_sym) + list(bc_extrapolate) + list(bc_inlet) + list(bc_outlet)
        ty = [BC_DIRICHLET] * len(bc_dirichlet) + [BC_SYMMETRY] * len(bc_sym) + \
             [BC_EXTRAPOLATE] * len(bc_extrapolate) + [BC_INLET] * len(bc_inlet) + \
             [BC_OUTLET] * len(bc_outlet)
        self._ss, pss = _i32(np.array(ss or [0], dtype=np.int32))
        self._ty, pty = _i32(np.array(ty or [1], dtype=np.int32))
        self.cfg = qdg_config(struct_size=C.sizeof(qdg_config), device=device, ndof=ndof,
                              rdof=ndof, flux=FLUX[flux], limiter=LIMITER[limiter],
                              problem=PROBLEM[problem], nbc=len(ss), bc_sideset=pss, bc_type=pty,
                              gamma=gamma, pstiff=pstiff, cv=cv, cweight=cweight, alpha=alpha,
                              beta=beta, p0=p0, cfl=cfl, dt=dt, pde=PDE[pde], pref=1 if pref else 0, tolref=tolref,
                              betax=betax, betay=betay, betaz=betaz, r0=r0, ce=ce, kappa=kappa,
                              ncomp=(nc if pde == "transport" else 0), tr_u0=psd[0], tr_lambda=psd[1],
                              tr_diffusivity=psd[2])
        self.h = C.c_void_p()
        _chk(L.qdg_ctx_create(C.byref(self.cfg), C.byref(self.h)))
        self.ndof = ndof
        self.ncomp = nc
        self.nprop = nc * ndof
        for k, v in {**default_options, **(options or {})}.items():
            self.set_option(k, v)

    def solution(self, pts, t):
        """Problem::solution at points [n,3] -> [n, ncomp]"""
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        n = pts.shape[0]
        x, y, z = (np.ascontiguousarray(pts[:, d]) for d in range(3))
        nc = self.nprop // self.ndof
        out = np.zeros((n, nc))
        _chk(lib().qdg_solution(self.h, C.c_size_t(n), x.ctypes.data_as(c_f64p), y.ctypes.data_as(c_f64p),
                                z.ctypes.data_as(c_f64p), C.c_double(t), out.ctypes.data_as(c_f64p)))
        return out

    def field_names(self):
        """DGPDE::fieldNames (+ 'ndof' with p-adaptive DG)"""
        n = C.c_size_t()
        _chk(lib().qdg_ctx_field_count(self.h, C.byref(n)))
        lib().qdg_ctx_field_name.restype = C.c_char_p
        lib().qdg_ctx_field_name.argtypes = [C.c_void_p, C.c_size_t]
        return [lib().qdg_ctx_field_name(self.h, i).decode() for i in range(n.value)]

    def field_output_from(self, t, geoElem, U):
        """stateless DGPDE::fieldOutput: [nfield, nunk]"""
        ge, pge = _f64(geoElem)
        U, pU = _f64(U)
        nunk = len(ge) // 4
        nf = len(self.field_names()) - (1 if self.cfg.pref else 0)
        out = np.zeros((nf, nunk))
        _chk(lib().qdg_field_output_from(self.h, C.c_double(t), C.c_size_t(nunk), pge, pU,
                                         out.ctypes.data_as(c_f64p)))
        return out

    def avg_elem_to_node(self, inpoel, nnode, U):
        """stateless DGPDE::avgElemToNode: [6, nnode]"""
        inp, pinp = _sz(np.asarray(inpoel).reshape(-1))
        U, pU = _f64(U)
        out = np.zeros((6, int(nnode)))
        _chk(lib().qdg_avg_elem_to_node(self.h, C.c_size_t(len(inp) // 4), C.c_size_t(int(nnode)), pinp, pU,
                                        out.ctypes.data_as(c_f64p)))
        return out

    def initialize_from(self, inpoel, coord, t=0.0, L=None, nielem=None):
        """DGPDE::initialize without an uploaded mesh (inpoel, coord, L only)"""
        inp = np.asarray(inpoel).reshape(-1, 4)
        nie = inp.shape[0] if nielem is None else int(nielem)
        inp, pinp = _sz(inp.reshape(-1))
        coord = np.asarray(coord, dtype=np.float64)
        x, px = _f64(coord[:, 0]); y, py = _f64(coord[:, 1]); z, pz = _f64(coord[:, 2])
        U = np.zeros((len(inp) // 4) * self.nprop)
        if L is not None:
            L, pL = _f64(L)
        else:
            pL = None
        _chk(lib().qdg_initialize_from(self.h, C.c_size_t(nie), C.c_size_t(coord.shape[0]), pinp, px, py, pz,
                                       pL, C.c_double(t), U.ctypes.data_as(c_f64p)))
        return U

    def device_memory(self):
        """(free, total, reserved by qdg_device_pool_reserve) bytes of this context's device"""
        f, t, r = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _chk(lib().qdg_device_memory(self.h, C.byref(f), C.byref(t), C.byref(r)))
        return f.value, t.value, r.value

    def reserve_device_memory(self, nbytes):
        """qdg_device_pool_reserve: one region from the driver now, the library's later allocations out of it"""
        _chk(lib().qdg_device_pool_reserve(self.h, C.c_size_t(int(nbytes))))

    def device_alloc(self, nbytes):
        """qdg_device_alloc: device pointer (int) to `nbytes` of this context's device, from the library's pool"""
        p = C.c_void_p()
        _chk(lib().qdg_device_alloc(self.h, C.c_size_t(int(nbytes)), C.byref(p)))
        return p.value or 0

    def device_free(self, ptr):
        _chk(lib().qdg_device_free(self.h, C.c_void_p(ptr)))

    def set_stream(self, stream_ptr):
        _chk(lib().qdg_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        _chk(lib().qdg_ctx_synchronize(self.h))

    def set_option(self, name, value):
        """qdg_ctx_set_option: tuning / A-B switch by name (see include/qdg.h); set before the
        meshes it concerns are created"""
        _chk(lib().qdg_ctx_set_option(self.h, name.encode(), C.c_int(int(value))))

    def get_option(self, name):
        v = C.c_int()
        _chk(lib().qdg_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def close(self):
        if self.h:
            lib().qdg_ctx_destroy(self.h)
            self.h = C.c_void_p()


class Mesh:
    """One uploaded mesh chunk (qdg_mesh).  Arguments are the reference-shaped
    arrays of one DG chare: inpoel[4*nunk], coord[nnode,3], FaceData members,
    geoFace[7*nfac], geoElem[4*nunk], bface."""

    def __init__(self, ctx, nielem, inpoel, coord, esuf, esuel, inpofa, geoFace, geoElem,
                 bface, nbfac, elem_gid=None):
        L = lib()
        self.ctx = ctx
        inp, pinp = _sz(np.asarray(inpoel).reshape(-1))
        self.nunk = len(inp) // 4
        self.nielem = int(nielem)
        coord = np.asarray(coord, dtype=np.float64)
        x, px = _f64(coord[:, 0]); y, py = _f64(coord[:, 1]); z, pz = _f64(coord[:, 2])
        esuf, pesuf = _i32(esuf); esuel, pesuel = _i32(esuel)
        inpofa, pinpofa = _sz(inpofa)
        gf, pgf = _f64(geoFace); ge, pge = _f64(geoElem)
        ids = sorted(bface)
        set_id, pid = _i32(np.array(ids or [0], dtype=np.int32))
        off = np.zeros(len(ids) + 1, dtype=np.uint64)
        faces = []
        for i, s_ in enumerate(ids):
            faces.append(np.asarray(bface[s_], dtype=np.uint64))
            off[i + 1] = off[i] + len(faces[-1])
        face = np.concatenate(faces) if faces else np.zeros(1, np.uint64)
        if face.size == 0:
            face = np.zeros(1, np.uint64)
        face, pface = _sz(face)
        off, poff = _sz(off)
        bf = qdg_bface(nset=len(ids), set_id=pid, set_off=poff, face=pface)
        self.h = C.c_void_p()
        pgid = None                 # global tet ids: faces oriented as in the serial run (qdg_mesh_upload_gid)
        if elem_gid is not None:
            gid, pgid = _sz(np.asarray(elem_gid))
            if len(gid) != self.nunk:
                raise QdgError("Mesh: elem_gid needs one entry per tet (ghosts included)")
        _chk(L.qdg_mesh_upload_gid(ctx.h, C.c_size_t(self.nielem), C.c_size_t(self.nunk),
                                   C.c_size_t(coord.shape[0]), pinp, px, py, pz, C.c_size_t(nbfac),
                                   C.c_size_t(len(esuf) // 2), pesuf, pesuel, pinpofa, pgf, pge,
                                   C.byref(bf), pgid, C.byref(self.h)))
        self.nprop = ctx.nprop

    # stateless DGPDE-shaped calls
    def lhs(self):
        Lm = np.zeros(self.nunk * self.nprop)
        _chk(lib().qdg_lhs(self.h, Lm.ctypes.data_as(c_f64p)))
        return Lm

    def initialize(self, t=0.0):
        U = np.zeros(self.nunk * self.nprop)
        _chk(lib().qdg_initialize(self.h, C.c_double(t), U.ctypes.data_as(c_f64p)))
        return U

    def rhs(self, t, U):
        U, pU = _f64(U)
        R = np.zeros(self.nunk * self.nprop)
        _chk(lib().qdg_rhs(self.h, C.c_double(t), pU, R.ctypes.data_as(c_f64p)))
        return R

    def dt(self, U):
        U, pU = _f64(U)
        v = C.c_double(0.0)
        _chk(lib().qdg_dt(self.h, pU, C.byref(v)))
        return v.value

    def limit(self, U):
        U = np.array(U, dtype=np.float64, copy=True)
        _chk(lib().qdg_limit(self.h, U.ctypes.data_as(c_f64p)))
        return U

    # resident path
    def state_upload(self, U):
        U, pU = _f64(U)
        _chk(lib().qdg_state_upload(self.h, pU))

    def state_download(self):
        U = np.zeros(self.nunk * self.nprop)
        _chk(lib().qdg_state_download(self.h, U.ctypes.data_as(c_f64p)))
        return U

    def state_initialize(self, t=0.0):
        _chk(lib().qdg_state_initialize(self.h, C.c_double(t)))

    def field_output(self, t=0.0):
        """Problem::fieldOutput of the resident state at time t: every field of the
        Problem's list [nfield, nielem] (+ ndof with p-adaptive DG) and the names"""
        n = C.c_size_t()
        _chk(lib().qdg_field_count(self.h, C.byref(n)))
        out = np.zeros((n.value, self.nielem))
        _chk(lib().qdg_field_output(self.h, C.c_double(t), out.ctypes.data_as(c_f64p)))
        lib().qdg_field_name.restype = C.c_char_p
        lib().qdg_field_name.argtypes = [C.c_void_p, C.c_size_t]
        return out, [lib().qdg_field_name(self.h, i).decode() for i in range(n.value)]

    def stage_pdg(self):
        _chk(lib().qdg_stage_pdg(self.h))

    def stage_pdg_eval(self):
        _chk(lib().qdg_stage_pdg_eval(self.h))

    def stage_pdg_propagate(self):
        _chk(lib().qdg_stage_pdg_propagate(self.h))

    def ndofel_get(self):
        a = np.zeros(self.nunk, dtype=np.uint64)
        _chk(lib().qdg_ndofel_get(self.h, a.ctypes.data_as(c_szp)))
        return a.astype(np.int64)

    def ndofel_set(self, ndofel):
        a, pa = _sz(ndofel)
        _chk(lib().qdg_ndofel_set(self.h, pa))

    def stage_limit(self):
        _chk(lib().qdg_stage_limit(self.h))

    def stage_dt(self, tleft=1e300):
        _chk(lib().qdg_stage_dt(self.h, C.c_double(tleft)))

    def stage_dt_get(self):
        v = C.c_double(0.0)
        _chk(lib().qdg_stage_dt_get(self.h, C.byref(v)))
        return v.value

    def stage_dt_set(self, dt):
        _chk(lib().qdg_stage_dt_set(self.h, C.c_double(dt)))

    def stage_dt_device_ptr(self):
        p = C.c_void_p()
        _chk(lib().qdg_stage_dt_device_ptr(self.h, C.byref(p)))
        return p.value

    def state_device_ptr(self):
        """(device pointer, row stride) of the resident state, for zero-copy plumbing"""
        p, st = C.c_void_p(), C.c_size_t()
        _chk(lib().qdg_state_device_ptr(self.h, C.byref(p), C.byref(st)))
        return p.value, st.value

    def stage_rhs_update(self, stage, t):
        _chk(lib().qdg_stage_rhs_update(self.h, C.c_int(stage), C.c_double(t)))

    def stage_rhs_dt(self, stage, t, tleft=1e300):
        _chk(lib().qdg_stage_rhs_dt(self.h, C.c_int(stage), C.c_double(t), C.c_double(tleft)))

    def stage_update(self, stage):
        _chk(lib().qdg_stage_update(self.h, C.c_int(stage)))

    def step(self, t, tleft=1e300, want_dt=True):
        v = C.c_double(0.0)
        _chk(lib().qdg_step(self.h, C.c_double(t), C.c_double(tleft),
                            C.byref(v) if want_dt else None))
        return v.value

    def diag(self, t_new):
        out = np.zeros(15)
        _chk(lib().qdg_diag(self.h, C.c_double(t_new), out.ctypes.data_as(c_f64p)))
        return out

    # halo
    def halo_setup(self, nbr_rank, send_lists, recv_counts):
        nbr, pn = _i32(np.array(list(nbr_rank) or [0], dtype=np.int32))
        soff = np.zeros(len(nbr_rank) + 1, dtype=np.uint64)
        roff = np.zeros(len(nbr_rank) + 1, dtype=np.uint64)
        for i, sl in enumerate(send_lists):
            soff[i + 1] = soff[i] + len(sl)
            roff[i + 1] = roff[i] + recv_counts[i]
        se = np.concatenate([np.asarray(s_, dtype=np.uint64) for s_ in send_lists]) \
            if len(send_lists) else np.zeros(1, np.uint64)
        if se.size == 0:
            se = np.zeros(1, np.uint64)
        se, pse = _sz(se); soff, psoff = _sz(soff); roff, proff = _sz(roff)
        _chk(lib().qdg_halo_setup(self.h, C.c_size_t(len(nbr_rank)), pn, psoff, pse, proff))
        self.send_off, self.recv_off = soff.astype(np.int64), roff.astype(np.int64)

    def halo_set_depth(self, nghost1):
        """qdg_halo_set_depth: the first nghost1 ghost rows are layer 1 of two ghost layers (the rank limits them
        itself, no exchange of the limited solution); 0: one layer"""
        _chk(lib().qdg_halo_set_depth(self.h, C.c_size_t(int(nghost1))))

    def halo_info(self):
        """-> (plan entries, layer-1 ghosts limited by this rank, packs folded into the producing kernels?)"""
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_int32()
        _chk(lib().qdg_halo_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, bool(c.value)

    def halo_buffers(self):
        a, b, r = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _chk(lib().qdg_halo_buffers(self.h, C.byref(a), C.byref(b), C.byref(r)))
        return a.value, b.value, r.value

    def halo_use_buffers(self, send_ptr, recv_ptr):
        _chk(lib().qdg_halo_use_buffers(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr)))

    def halo_sizes(self):
        a, b = C.c_size_t(), C.c_size_t()
        _chk(lib().qdg_halo_sizes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stage_dt_use_buffer(self, ptr):
        _chk(lib().qdg_stage_dt_use_buffer(self.h, C.c_void_p(ptr)))

    def state_rows_get(self, rows, packed_dev):
        """rows (caller's numbering) of the resident state -> packed device buffer (address)"""
        r, pr = _sz(np.asarray(rows))
        _chk(lib().qdg_state_rows_get(self.h, C.c_size_t(len(r)), pr, C.c_void_p(int(packed_dev))))

    def state_rows_put(self, rows, packed_dev):
        r, pr = _sz(np.asarray(rows))
        _chk(lib().qdg_state_rows_put(self.h, C.c_size_t(len(r)), pr, C.c_void_p(int(packed_dev))))

    def refine_uniform(self, host_copy=True):
        """qdg_mesh_refine_uniform: the re-mesh of this resident chunk on the device (needs context option
        keep_connectivity = 1 at build time).  Returns (new Mesh with the state handed over, Refined or None);
        this mesh stays valid until closed."""
        new = Mesh.__new__(Mesh)
        new.ctx, new.nprop, new.h = self.ctx, self.nprop, C.c_void_p()
        r = C.c_void_p()
        _chk(lib().qdg_mesh_refine_uniform(self.h, C.byref(new.h), C.byref(r) if host_copy else None))
        new.nielem = new.nunk = 8 * self.nielem
        return new, (Refined(r) if host_copy else None)

    def derefine_uniform(self, policy="first_child"):
        """qdg_mesh_derefine_uniform: the 8:1 coarsening of this resident chunk (no ghosts; its kept connectivity must
        be a uniform refinement in the library's order) on the device.  policy "first_child": a parent takes its first
        child's row (the inverse of the reference's row copy child <- parent); "mean": the volume-weighted mean of
        the children's means, higher-order DOFs zero.  Returns the new Mesh; this mesh stays valid until closed."""
        new = Mesh.__new__(Mesh)
        new.ctx, new.nprop, new.h = self.ctx, self.nprop, C.c_void_p()
        _chk(lib().qdg_mesh_derefine_uniform(self.h, C.c_int({"first_child": 0, "mean": 1}[policy]), C.byref(new.h)))
        new.nielem = new.nunk = self.nielem // 8
        return new

    def refine_chunk(self, nbr_rank=None, copy_mesh=False):
        """qdg_mesh_refine_chunk: the re-mesh of this rank's chunk WITH its ghost layer(s) on the device (built by
        mesh_from_connectivity(..., nielem, elem_gid) under keep_connectivity = 1, halo_setup [+ halo_set_depth]
        done).  Returns (new Mesh -- halo plan set up, owned state handed over --, chunk dict: nielem, gid, parent,
        nbr_rank, nbr_layer, nghost1, depth, send_lists, recv_counts [+ inpoel, coord, sidesets with copy_mesh]).
        (nbr_rank is ignored: the refined chunk's plan entries come back from the library.)"""
        L = lib()
        new = Mesh.__new__(Mesh)
        new.ctx, new.nprop, new.h = self.ctx, self.nprop, C.c_void_p()
        h = C.c_void_p()
        _chk(L.qdg_mesh_refine_chunk(self.h, C.byref(new.h), C.byref(h), C.c_int(1 if copy_mesh else 0)))
        try:
            n = [C.c_size_t() for _ in range(5)]
            _chk(L.qdg_chunk_refined_sizes(h, *[C.byref(v) for v in n]))
            nie2, nunk2, nn2, ntri2, nsend = (int(v.value) for v in n)
            ne_, ng1_ = C.c_size_t(), C.c_size_t()
            _chk(L.qdg_chunk_refined_plan(h, C.byref(ne_), C.byref(ng1_), None, None))
            nnbr, nghost1 = int(ne_.value), int(ng1_.value)
            nb2 = np.zeros(max(1, nnbr), dtype=np.int32); nl2 = np.ones(max(1, nnbr), dtype=np.int32)
            _chk(L.qdg_chunk_refined_plan(h, None, None, nb2.ctypes.data_as(c_i32p), nl2.ctypes.data_as(c_i32p)))
            gid = np.empty(nunk2, dtype=np.uint64); par = np.empty(nunk2, dtype=np.uint64)
            soff = np.zeros(nnbr + 1, dtype=np.uint64); slist = np.zeros(max(1, nsend), dtype=np.uint64)
            rc = np.zeros(max(1, nnbr), dtype=np.uint64)
            inp = np.empty(4 * nunk2 if copy_mesh else 1, dtype=np.uint64)
            c = np.empty((3, nn2 if copy_mesh else 1))
            tri = np.zeros(max(1, 3 * ntri2), dtype=np.uint64); tset = np.zeros(max(1, ntri2), dtype=np.int32)
            _chk(L.qdg_chunk_refined_get(h, inp.ctypes.data_as(c_szp) if copy_mesh else None, gid.ctypes.data_as(c_szp),
                                         par.ctypes.data_as(c_szp),
                                         c[0].ctypes.data_as(c_f64p) if copy_mesh else None,
                                         c[1].ctypes.data_as(c_f64p) if copy_mesh else None,
                                         c[2].ctypes.data_as(c_f64p) if copy_mesh else None,
                                         tri.ctypes.data_as(c_szp) if copy_mesh else None,
                                         tset.ctypes.data_as(c_i32p) if copy_mesh else None,
                                         soff.ctypes.data_as(c_szp), slist.ctypes.data_as(c_szp),
                                         rc.ctypes.data_as(c_szp)))
        except BaseException:
            new.close()                      # the new device mesh is not handed out: release it
            raise
        finally:
            L.qdg_chunk_refined_destroy(h)
        new.nielem, new.nunk = nie2, nunk2
        soff = soff.astype(np.int64)
        layers = [int(v) for v in nl2[:nnbr]]
        ch = {"nielem": nie2, "gid": gid.view(np.int64), "parent": par.view(np.int64),
              "nbr_rank": [int(v) for v in nb2[:nnbr]], "nbr_layer": layers, "nghost1": nghost1,
              "depth": 2 if 2 in layers else 1,
              "send_lists": [slist[soff[i]:soff[i + 1]].astype(np.int64) for i in range(nnbr)],
              "recv_counts": [int(v) for v in rc[:nnbr]]}
        if copy_mesh:
            tri = tri[:3 * ntri2].view(np.int64).reshape(-1, 3); tset = tset[:ntri2]
            ch.update(inpoel=inp.view(np.int64).reshape(-1, 4), coord=np.ascontiguousarray(c.T),
                      sidesets={int(s_): tri[tset == s_] for s_ in np.unique(tset)})
        return new, ch

    def profile_enable(self, on=True):
        _chk(lib().qdg_profile_enable(self.h, C.c_int(1 if on else 0)))

    def profile_read(self):
        n, ms = C.c_size_t(), C.c_double()
        _chk(lib().qdg_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def layout_stats(self):
        """-> dict: face tasks of the tile layout (in_tile, to_other_tiles, boundary, tiles)"""
        n = (C.c_size_t * 4)()
        _chk(lib().qdg_mesh_layout_stats(self.h, n))
        return {"in_tile": int(n[0]), "to_other_tiles": int(n[1]), "boundary": int(n[2]), "tiles": int(n[3])}

    def profile_read_all(self):
        """-> {"rhs": (launches, ms), "halo": (exchanges, ms), "allreduce": (calls, ms)} since the last read"""
        n, ms = (C.c_size_t * 3)(), (C.c_double * 3)()
        _chk(lib().qdg_profile_read_all(self.h, n, ms))
        return {k: (int(n[i]), float(ms[i])) for i, k in enumerate(("rhs", "halo", "allreduce"))}

    def step_graph_status(self):
        """-> (state, graphs, replays, error text): qdg_step_comm's hipGraph replay (option graph_step)"""
        st, ng, nr = C.c_int32(), C.c_int32(), C.c_int64()
        buf = C.create_string_buffer(512)
        _chk(lib().qdg_step_graph_status(self.h, C.byref(st), C.byref(ng), C.byref(nr), buf, C.c_size_t(512)))
        return st.value, ng.value, nr.value, buf.value.decode()

    def rhs_algorithmic_bytes(self):
        b = C.c_double()
        _chk(lib().qdg_rhs_algorithmic_bytes(self.h, C.byref(b)))
        return b.value

    def halo_copy_from(self, dst_row0, src, src_row0, nrows):
        """send-slab rows of `src` -> receive-slab rows of this chunk (same context)"""
        _chk(lib().qdg_halo_copy(self.h, C.c_size_t(int(dst_row0)), src.h, C.c_size_t(int(src_row0)),
                                 C.c_size_t(int(nrows))))

    def halo_pack(self):
        _chk(lib().qdg_halo_pack(self.h))

    def halo_exchange(self, comm):
        _chk(lib().qdg_halo_exchange(self.h, comm.h))

    def stage_dt_allreduce(self, comm):
        _chk(lib().qdg_stage_dt_allreduce(self.h, comm.h))

    def step_comm(self, comm, t, tleft=1e300, want_dt=False):
        v = C.c_double(0.0)
        _chk(lib().qdg_step_comm(self.h, comm.h, C.c_double(t), C.c_double(tleft),
                                 C.byref(v) if want_dt else None))
        return v.value

    def halo_unpack(self):
        _chk(lib().qdg_halo_unpack(self.h))

    def close(self):
        if self.h:
            lib().qdg_mesh_destroy(self.h)
            self.h = C.c_void_p()


def mesh_from_connectivity(ctx, inpoel, coord, sidesets, nielem=None, elem_gid=None):
    """qdg_mesh_from_connectivity / qdg_mesh_from_chunk[_gid]: a Mesh handle straight from
    (inpoel, coord, {side set id: triangles}); FaceData, geometry and the device layout are
    made on the GPU.  nielem < number of tets: the trailing tets are the chunk's ghost layer.
    elem_gid: global tet ids -- faces oriented by global id (a partitioned run = the serial run)."""
    inpoel = np.ascontiguousarray(inpoel, dtype=np.uint64).reshape(-1)
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    ids = sorted(sidesets)
    tri = np.concatenate([np.asarray(sidesets[s_], dtype=np.uint64).reshape(-1, 3) for s_ in ids]) \
        if ids else np.zeros((0, 3), dtype=np.uint64)
    tset = np.concatenate([np.full(len(sidesets[s_]), s_, np.int32) for s_ in ids]) \
        if ids else np.zeros(0, np.int32)
    tri = np.ascontiguousarray(tri.reshape(-1) if tri.size else np.zeros(3, np.uint64))
    tset = np.ascontiguousarray(tset if tset.size else np.zeros(1, np.int32))
    x, y, z = (np.ascontiguousarray(coord[:, d]) for d in range(3))
    m = Mesh.__new__(Mesh)
    m.ctx, m.nunk, m.nprop = ctx, inpoel.size // 4, ctx.nprop
    m.nielem = m.nunk if nielem is None else int(nielem)
    m.h = C.c_void_p()
    ntri = sum(len(sidesets[s_]) for s_ in ids)
    pgid = None
    if elem_gid is not None:
        gid = np.ascontiguousarray(elem_gid, dtype=np.uint64)
        if gid.size != m.nunk:
            raise QdgError("mesh_from_connectivity: elem_gid needs one entry per tet (ghosts included)")
        pgid = gid.ctypes.data_as(c_szp)
    _chk(lib().qdg_mesh_from_chunk_gid(ctx.h, C.c_size_t(m.nielem), C.c_size_t(m.nunk), C.c_size_t(coord.shape[0]),
                                       inpoel.ctypes.data_as(c_szp), x.ctypes.data_as(c_f64p),
                                       y.ctypes.data_as(c_f64p), z.ctypes.data_as(c_f64p),
                                       C.c_size_t(ntri), tri.ctypes.data_as(c_szp),
                                       tset.ctypes.data_as(c_i32p), pgid, C.byref(m.h)))
    return m


class Refined:
    """qdg_refined handle of qdg_mesh_refine_uniform: the refined mesh on the host, copied from the device by a
    second thread; get() waits for it.  -> coord[nnode, 3], inpoel[ne, 4], {side set: triangles}, parent[ne]"""

    def __init__(self, h):
        self.h = h

    def get(self):
        L = lib()
        ne, nn, nt = C.c_size_t(), C.c_size_t(), C.c_size_t()
        _chk(L.qdg_refined_sizes(self.h, C.byref(ne), C.byref(nn), C.byref(nt)))
        ne, nn, nt = ne.value, nn.value, nt.value
        inp = np.empty(4 * ne, dtype=np.uint64); par = np.empty(ne, dtype=np.uint64)
        c = np.empty((3, nn)); tri = np.zeros(max(1, 3 * nt), dtype=np.uint64)
        tset = np.zeros(max(1, nt), dtype=np.int32)
        _chk(L.qdg_refined_get(self.h, None, inp.ctypes.data_as(c_szp), par.ctypes.data_as(c_szp),
                               c[0].ctypes.data_as(c_f64p), c[1].ctypes.data_as(c_f64p),
                               c[2].ctypes.data_as(c_f64p), tri.ctypes.data_as(c_szp)))
        if nt:
            _chk(L.qdg_refined_tri_sets(self.h, tset.ctypes.data_as(c_i32p)))
        tri = tri[:3 * nt].view(np.int64).reshape(-1, 3)
        tset = tset[:nt]
        ss = {int(s_): tri[tset == s_] for s_ in np.unique(tset)}
        return np.ascontiguousarray(c.T), inp.view(np.int64).reshape(-1, 4), ss, par.view(np.int64)

    def close(self):
        if self.h:
            lib().qdg_refined_destroy(self.h)
            self.h = C.c_void_p()


def dev_facedata(ctx, inpoel, coord, triinpoel):
    """FaceData arrays and geometry of one chunk generated on the GPU
    (qdg_dev_facedata): dict with esuel, nipfac, inpofa, esuf, belem, geoFace, geoElem."""
    inpoel = np.ascontiguousarray(inpoel, dtype=np.uint64).reshape(-1)
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    tri = np.ascontiguousarray(triinpoel, dtype=np.uint64).reshape(-1)
    nelem, nnode, nbfac = inpoel.size // 4, coord.shape[0], tri.size // 3
    nfmax = nbfac + 2 * nelem
    x, y, z = (np.ascontiguousarray(coord[:, d]) for d in range(3))
    esuel = np.zeros(4 * nelem, dtype=np.int32)
    inpofa = np.zeros(3 * nfmax, dtype=np.uint64)
    esuf = np.zeros(2 * nfmax, dtype=np.int32)
    belem = np.zeros(max(1, nbfac), dtype=np.uint64)
    geoFace = np.zeros(7 * nfmax)
    geoElem = np.zeros(4 * nelem)
    nip = C.c_size_t()
    trip = tri if nbfac else np.zeros(3, dtype=np.uint64)
    _chk(lib().qdg_dev_facedata(ctx.h, C.c_size_t(nelem), C.c_size_t(nnode), inpoel.ctypes.data_as(c_szp),
                                x.ctypes.data_as(c_f64p), y.ctypes.data_as(c_f64p), z.ctypes.data_as(c_f64p),
                                C.c_size_t(nbfac), trip.ctypes.data_as(c_szp), esuel.ctypes.data_as(c_i32p),
                                C.byref(nip), inpofa.ctypes.data_as(c_szp), esuf.ctypes.data_as(c_i32p),
                                belem.ctypes.data_as(c_szp), geoFace.ctypes.data_as(c_f64p),
                                geoElem.ctypes.data_as(c_f64p)))
    n = nip.value
    return dict(esuel=esuel, nipfac=n, inpofa=inpofa[:3 * n], esuf=esuf[:2 * n], belem=belem[:nbfac],
                geoFace=geoFace[:7 * n], geoElem=geoElem)


class Comm:
    """RCCL communicator of libqdg (qdg_comm_*): one rank per GPU."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        _chk(lib().qdg_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, nranks, rank, unique_id):
        assert len(unique_id) == 128
        self.h = C.c_void_p()
        self.nranks, self.rank = nranks, rank
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _chk(lib().qdg_comm_create(ctx.h, C.c_int(nranks), C.c_int(rank), buf, C.byref(self.h)))

    def info(self):
        """(ranks, rank, device) as RCCL reports them (ncclCommCount / UserRank / CuDevice)"""
        n, r, d = C.c_int(), C.c_int(), C.c_int()
        _chk(lib().qdg_comm_info(self.h, C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value

    def close(self):
        if self.h:
            _chk(lib().qdg_comm_destroy(self.h))
            self.h = C.c_void_p()
