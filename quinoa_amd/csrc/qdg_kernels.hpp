// qdg_kernels.hpp -- host-callable launchers of the gfx950 kernels
// (definitions in qdg_kernels.hip, qdg_rhs_p1.hip, qdg_rhs_p2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "qdg_device.hpp"

namespace qdg {

hipError_t upload_tables(const Tables<1>& t1, const Tables<4>& t4, const Tables<10>& t10,
                         const QuadTet* qinit, const QuadTet* qdiag);

void launch_task_geo(size_t nslot, const int* task_a, const int* task_f, const double* fgeo, double* tgeo,
                     hipStream_t s);
void launch_rhs(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                hipStream_t s);
void launch_rhs_dt(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                   double* blockmin, double scale, double tleft, double* out_raw, double* out_dt,
                   hipStream_t s);
void launch_rhs_rk(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                   double a, double b, const double* dt, const double* Un, hipStream_t s);
void launch_dt_final(const double* blockmin, int n, double scale, double tleft, double* out_raw,
                     double* out_dt, hipStream_t s);
// DG-P2 (qdg_rhs_p2.hip): mode 0 R = rhs(U); 1 + per-workgroup min(vol/delt) into blockmin
// [p2_rhs_blocks(m)]; 2 SSP-RK3 update fused in
int p2_rhs_blocks(const DevMesh& m);
void launch_rhs_p2(const DevMesh& m, const Phys& ph, double t, const double* U, double* R, int mode,
                   double* blockmin, double a, double b, const double* dt, const double* Un, hipStream_t s);
void launch_rhs_p1(const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                   bool with_dt, double* blockmin, double scale, double tleft, double* out_raw,
                   double* out_dt, hipStream_t s);
// range launches: workgroup tiles [first, first+count), count < 0 = all the rest
void launch_rhs_p1t(const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                    bool with_dt, double* blockmin, double scale, double tleft, double* out_raw,
                    double* out_dt, hipStream_t s, int first = 0, int count = -1);
void launch_rhs_p1t_rk(const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                       double a, double b, const double* dt, const double* Un, hipStream_t s,
                       int first = 0, int count = -1);
void launch_superbee(int ndof, const DevMesh& m, double* U, hipStream_t s, int first = 0,
                     int count = -1);
// stage-0 RK update fused with the Superbee limiter of stage 1 (DG-P1), and its halo pack
void launch_upd_superbee(const DevMesh& m, const double* dt, const double* U0, const double* R,
                         double* Uout, hipStream_t s);
void launch_halo_pack_upd(const double* U0, const double* R, const double* dt, const double* vol,
                          const int* send_elem, int nsend, double* slab, hipStream_t s);
void launch_weno(int ndof, const DevMesh& m, double cweight, const double* Uin, double* Uout,
                 hipStream_t s);
void launch_copy_planes(const double* src, double* dst, int nprop, int n, int stride,
                        hipStream_t s);
int dt_blocks(const DevMesh& m);
void launch_dt(int ndof, const DevMesh& m, const Phys& ph, const double* U, double* blockmin,
               double scale, double tleft, double* out_raw, double* out_dt, hipStream_t s);
void launch_rk(int ndof, const DevMesh& m, double a, double b, const double* dt, const double* Un,
               const double* R, const double* U, double* Uout, hipStream_t s);
void launch_rhs_p1_rk(const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                      double a, double b, const double* dt, const double* Un, hipStream_t s);
void launch_mass(int ndof, const DevMesh& m, double* L, hipStream_t s);
void launch_init(int ndof, const DevMesh& m, const Phys& ph, double t, double* U, hipStream_t s);
void launch_diag(int ndof, const DevMesh& m, const Phys& ph, double t_new, const double* U,
                 double* part, double* out, hipStream_t s);
void launch_aos2soa(const double* aos, int nprop, const int* d2h, int n0, int n1, int stride,
                    double* soa, hipStream_t s);
void launch_soa2aos(const double* soa, int nprop, const int* d2h, int n0, int n1, int stride,
                    double* aos, hipStream_t s);
// ndofel != null (p-adaptive DG): slab rows are nprop + 1 doubles, the last one the tet's ndof
void launch_halo_pack(const double* U, int nprop, int stride, const int* send_elem, int nsend,
                      double* slab, hipStream_t s, const int* ndofel = nullptr);
void launch_halo_unpack(const double* slab, int nprop, int stride, int nie, int nrecv, double* U,
                        hipStream_t s, int* ndofel = nullptr);

void launch_state_transfer(int nrow, int nprop, const int* d2h_to, const int* parent, const int* h2d_from,
                           const double* Ufrom, double* Uto, hipStream_t s);
void launch_rows_gather(size_t n, int nprop, const int* drow, const double* U, double* packed, hipStream_t s);
void launch_rows_scatter(size_t n, int nprop, const int* drow, const double* packed, double* U, hipStream_t s);
void launch_solution(int pde, int ncomp, const Phys& ph, int n, const double* x, const double* y, const double* z,
                     double t, double* out, hipStream_t s);
// number of element fields of Problem::fieldNames (without the ndof column of p-adaptive runs)
int field_count(int pde, int ncomp, int problem);
void launch_field_output(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U,
                         const double* geoElem, int nrows, double* out, hipStream_t s);
void launch_avg_elem_to_node(const Phys& ph, int rdof, int nelem, int nnode, const int* inpoel,
                             const double* U, double* out, double* count, hipStream_t s);
void launch_tet_volumes(int nelem, int stride, const int* inpoel, const double* x, const double* y,
                        const double* z, double* vol, hipStream_t s);
// p-adaptive DG (DG::eval_ndof, propagate_ndof, zeroing of P0 high-order DOFs)
void launch_pdg_eval(const DevMesh& m, const double* U, double tolref, int* ndofel, hipStream_t s);
void launch_pdg_propagate(const DevMesh& m, const int* in, int* out, hipStream_t s);
void launch_pdg_zero(const DevMesh& m, const int* ndofel, double* U, hipStream_t s);
void launch_fill_int(int* p, int n, int v, hipStream_t s);

}  // namespace qdg
