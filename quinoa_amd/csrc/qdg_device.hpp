// qdg_device.hpp -- data structures shared by the host layer and the gfx950
// kernels of the DG compressible-flow path (MI355X / CDNA4 only).
//
// HBM layout
//
//   U, Un, R   : element-major rows in DEVICE element order,
//                  U[e*NPROP + c*NDOF + k],  NPROP = 5*NDOF
//                (the reference's tk::Fields row, src/Base/Data.hpp:462-471).
//                A face-neighbour gather reads one contiguous row (P1: 160 B,
//                1.25 cache lines) with 16-byte loads; measured fabric traffic
//                of the P1 RHS is 1.7x lower than with per-DOF planes, where
//                the same gather touches 20 different lines.
//   integer connectivity and per-element scalars are struct-of-arrays planes
//   (they are only ever read by their own element, fully coalesced):
//   inpoel     : 4 planes of int32 (device-order elements, renumbered nodes)
//   nbr        : 4 planes of int32, neighbour across local face lf:
//                  >= 0  device id of the neighbour (ghosts: >= nie)
//                  <  0  physical boundary, -(1+bc), bc in {0 none, QDG_BC_*}
//   finfo      : 4 planes of int32; bits 0-5: local node ids (2 bits each) in
//                the NEIGHBOUR of this face's 3 nodes, taken in this element's
//                lpofa[lf] order; bit 6: 1 if this element is the face's left
//                element (esuf[2f]), i.e. the stored normal points outward
//   fid        : 4 planes of int32, device face id -> farea/fnx/fny/fnz
//   vol        : tet volume (geoElem(e,0)), x/y/z node coordinates
//
// Device order: interior tets are sorted along a Morton curve of their
// centroids (neighbour gathers hit L2), nodes are renumbered by first touch,
// faces are enumerated in device-element order; ghosts keep rows [nie, ne).
#pragma once
#include <cstddef>
#include <cstdint>

namespace qdg {

constexpr int NCOMP = 5;
constexpr int FOLD_SLOTS = 8;     // slab rows a tet can appear in (one per (neighbour rank, ghost layer) entry that lists it)

constexpr int ngfa(int ndof)   { return ndof == 1 ? 1 : ndof == 4 ? 3 : 6; }
constexpr int ngvol(int ndof)  { return ndof == 1 ? 1 : ndof == 4 ? 5 : 11; }
constexpr int ngdiag(int ndof) { return ndof == 1 ? 1 : ndof == 4 ? 4 : 14; }
constexpr int nginit(int ndof) { return ndof == 1 ? 1 : 14; }

struct DevMesh {
  int nie;          // interior elements
  int ne;           // interior + ghosts
  int stride;       // plane stride (ne rounded up to 64)
  int nnode;
  int nfac;         // device faces
  const int* inpoel;   // [4][stride]
  const int* nbr;      // [4][stride]
  const int* finfo;    // [4][stride]
  const int* fid;      // [4][stride]
  const double* x;
  const double* y;
  const double* z;
  const double* farea;
  const double* fnx;
  const double* fny;
  const double* fnz;
  const double* vol;   // [stride]
  const double* fgeo;  // [nfac][4] packed {area, nx, ny, nz}: one 32-byte record per face
  const double* xyz4;  // [nnode][4] packed {x, y, z, 0}: one 32-byte record per node
  const int* d2h;      // [ne] device row -> host row
  // face tasks of the tile kernel (qdg_kernels.hip: k_rhs_p1t): the interior tets
  // are cut into tiles of TILE consecutive device rows; every face of a tile is
  // listed ONCE (by its left tet when both tets are in the tile)
  int ntile;
  int ntile_inner;     // tiles that end at or before row ninner (see blk0 below)
  const int* tile_row; // [ntile+1] first device row of each tile (<= TILE rows per tile)
  int tile_rows;       // > 0: every tile has this many rows (tile t starts at t*tile_rows), no table look-up
  const int* tile_off; // [ntile+1] first task of each tile (task_stride == 0)
  int task_stride;     // > 0: padded task lists, tile t owns slots [t*stride, (t+1)*stride), unused = -1
  const int* task_a;   // packed: e_local(8) lf(2) own_left(1) code(6) kind(2) bc(2) partner_local(8)
  const int* task_nb;  // neighbour device row (kind EXT), else 0
  const int* task_f;   // device face id
  const double* tgeo;  // [task slot][4] the face's {area, nx, ny, nz} in TASK order (padded lists only,
                       // else null): one coalesced 32-byte record per lane instead of a gather by face id
  // range launches (halo overlap): first workgroup-tile of this launch.  Device
  // rows [0, ninner) are tets without a ghost neighbour, [ninner, nie) the tets
  // next to the halo, so a launch over the leading tiles never reads a ghost row.
  int blk0;
  int ninner;
  // halo pack folded into the producing kernels (set by qdg_step_comm for its own launches only): a row
  // d >= ninner that neighbours need is ALSO written to the send slab, at up to FOLD_SLOTS slab rows
  // fold_slot[FOLD_SLOTS * (d - ninner) ...] (-1 = none); null: no folding
  const int* fold_slot;
  double* fold_slab;
  // limiter range: rows [0, nlim).  nlim = nie, or nie + the chunk's layer-1 ghosts when the rank limits them
  // itself (two ghost layers, qdg_halo_set_depth: their nbr rows are filled); 0 reads as nie
  int nlim;
  int lim_write_all;   // 1: the Superbee kernel writes every tile back, changed or not (context option
                       // "limiter_write_all": the flow-independent lower bound of the limiter pass, for measurement)
  int ncomp;        // 5: CompFlow; dg::Transport: its number of scalars (rows of ncomp*ndof doubles)
  int pde;          // 0: CompFlow, 1: dg::Transport (QDG_PDE_*)
  // p-adaptive DG (scheme pdg): DG::m_ndof per device row, 1 or 4; null otherwise
  const int* ndofel;
};

#ifndef QDG_TILE
#define QDG_TILE 248
#endif
#ifndef QDG_TILE_BS
#define QDG_TILE_BS 256
#endif
constexpr int TILE = QDG_TILE;
constexpr int TILE_BS = QDG_TILE_BS;   // workgroup size of the tile kernel
static_assert(TILE <= TILE_BS && TILE <= 512, "one lane per tet; local ids are 8 or 9 bits");
// packed task word: e_local(TLB) lf(2) own_left(1) code(6) kind(2) bc(2) partner_local(TLB); TLB = 8
// bits for tiles of up to 256 rows, 9 beyond (31 bits: the sign stays free for "unused slot")
constexpr int TLB = TILE > 256 ? 9 : 8;
#define TASK_PACK(el, lf, own_left, code, kind, bc, pl)                                                   \
  ((int)(el) | ((lf) << qdg::TLB) | ((own_left) << (qdg::TLB + 2)) | ((code) << (qdg::TLB + 3)) |          \
   ((kind) << (qdg::TLB + 9)) | ((bc) << (qdg::TLB + 11)) | ((pl) << (qdg::TLB + 13)))
#define TASK_EL(a) ((a) & ((1 << qdg::TLB) - 1))
#define TASK_LF(a) (((a) >> qdg::TLB) & 3)
#define TASK_OWNLEFT(a) (((a) >> (qdg::TLB + 2)) & 1)
#define TASK_CODE(a) (((a) >> (qdg::TLB + 3)) & 63)
#define TASK_KIND(a) (((a) >> (qdg::TLB + 9)) & 3)
#define TASK_BC(a) (((a) >> (qdg::TLB + 11)) & 3)
#define TASK_PL(a) (((a) >> (qdg::TLB + 13)) & ((1 << qdg::TLB) - 1))
//   // tets per tile: 2 x (rows + accumulators + dt sums) fit the CU's 160 KiB LDS
enum { TASK_INT = 0, TASK_EXT = 1, TASK_BND = 2 };

struct Phys {
  double gamma, pstiff, cweight, cv;
  double alpha, beta, p0;
  double betax, betay, betaz, r0, ce, kappa;   // nl_energy_growth
  int flux, problem, limiter;
  // shear_diff (dg::Transport), per scalar: u0, lambda[2], diffusivity[3]
  double sd_u0[5], sd_lambda[10], sd_diff[15];
};

// Quadrature / basis tables for one polynomial order, kept in constant memory
// so that loop-uniform indices turn into scalar loads.
template <int NDOF> struct Tables {
  static constexpr int NGF = ngfa(NDOF);
  static constexpr int NGV = ngvol(NDOF);
  double fw[NGF];             // triangle weights          (Quadrature.cpp:261-339)
  double fs[NGF][3];          // barycentric weights of the face's 3 nodes
  double fB[4][NGF][NDOF];    // own basis at the Gauss points of local face lf
  double vw[NGV];             // tet weights               (Quadrature.cpp:16-259)
  double vc[NGV][3];          // tet Gauss point reference coordinates
  double vB[NGV][NDOF];       // basis at the volume Gauss points
  double vdB[NGV][3][NDOF];   // dB_k/dxi_j at the volume Gauss points
};

// Tables of the two-lanes-per-tet DG-P2 kernel (k_rhs_p2s): lane half h of a tet works on the
// modes k in [5h, 5h+5), so every basis value it needs is indexed by (h, point) per lane and
// comes from LDS instead of scalar constants.
struct P2Split {
  // basis at the six face points of a tet whose local nodes (m0, m1, m2) carry the face's three
  // nodes: [rank of (m0, m1, m2)][point][h][k - 5h], 5 values padded to 6 (48-byte records)
  double face[24][6][2][6];
  double fq[6][4];           // face point: barycentric weights of the 3 face nodes, weight
  // volume points: [point][h]{ B[5], dB/dxi[5], dB/deta[5], dB/dzeta[5] }; point 11 = point 0
  // with weight 0 (the pair of lanes takes the 11 points two at a time)
  double vol[12][2][20];
  double vw[12];
  double vc[12][4];
};

// generic tet rule (initialisation: 14 points; diagnostics: 1/4/14 points)
struct QuadTet {
  int ng;
  double c[14][3];
  double w[14];
};

}  // namespace qdg
