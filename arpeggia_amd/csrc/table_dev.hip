// The contact table on the device: SURVEY.md 8f rows f1 (ring / side-chain plane fits, ring-atom and ring-ring rows,
// complex.rs:301-405, residues.rs:270-298, aromatic.rs:14-64) and f2 (row expansion, the 10-key sort of mod.rs:120-134, the
// side-chain plane statistics of complex.rs:137-174).  The host (table.cpp) only keeps the bookkeeping: which entities exist and
// their strings.  Written for gfx950; sorting and scans go through hipCUB (library primitives, not the hot path).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "arp_internal.h"
#include "host_common.h"
#include "table_dev.h"

namespace arp {

#define TRY_HIP(expr)                                                                                                     \
    do {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                           \
        if (e_ != hipSuccess) {                                                                                           \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__, #expr);       \
            return (e_ == hipErrorOutOfMemory) ? ARP_ERR_OOM : ARP_ERR_HIP;                                               \
        }                                                                                                                 \
    } while (0)

struct PlaneD { double c[3], n[3]; };

// ---- plane fits (residues.rs:270-298): centroid + direction of least variance --------------------------------------
// The same cyclic Jacobi iteration as the host's former fit (table.cpp fit_plane): nalgebra's svd.u.column(2) is that direction up
// to sign, and every use folds the angle into [0, 90] degrees.
__device__ bool fit_plane_dev(const double *x, const double *y, const double *z, const uint32_t *idx, uint32_t lo, uint32_t hi, const uint8_t *bits, uint8_t want,
                              PlaneD *out) {
    uint32_t n = 0;
    double c[3] = {0, 0, 0};
    for (uint32_t p = lo; p < hi; p++) {
        const uint32_t a = idx[p];
        if (!(bits[a] & want)) continue;
        c[0] += x[a]; c[1] += y[a]; c[2] += z[a];
        n++;
    }
    if (n < 3) return false;  // residues.rs:273
    for (int k = 0; k < 3; k++) c[k] /= (double)n;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (uint32_t p = lo; p < hi; p++) {
        const uint32_t a = idx[p];
        if (!(bits[a] & want)) continue;
        const double d[3] = {x[a] - c[0], y[a] - c[1], z[a] - c[2]};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] += d[i] * d[j];
    }
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; sweep++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                if (fabs(A[p][q]) <= 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq; }
                for (int k = 0; k < 3; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk; }
                for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq; }
            }
    }
    const double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
    double v[3];
    if (e1 < e0 && e1 <= e2) { v[0] = V[0][1]; v[1] = V[1][1]; v[2] = V[2][1]; }
    else if (e2 < e0 && e2 < e1) { v[0] = V[0][2]; v[1] = V[1][2]; v[2] = V[2][2]; }
    else { v[0] = V[0][0]; v[1] = V[1][0]; v[2] = V[2][0]; }
    const double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int k = 0; k < 3; k++) { out->c[k] = c[k]; out->n[k] = v[k] / nn; }
    return true;
}

__global__ __launch_bounds__(128) void k_fit_planes(uint32_t n_res, const uint32_t *res_atom_ptr, const uint32_t *res_atom_idx, const uint8_t *bits, const double *x,
                                                    const double *y, const double *z, PlaneD *ring, PlaneD *sc, uint8_t *valid) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_res) return;
    const uint32_t lo = res_atom_ptr[r], hi = res_atom_ptr[r + 1];
    uint8_t any = 0;
    for (uint32_t p = lo; p < hi; p++) any |= bits[res_atom_idx[p]];
    uint8_t v = 0;
    if ((any & 1u) && fit_plane_dev(x, y, z, res_atom_idx, lo, hi, bits, 1u, &ring[r])) v |= 1u;
    if ((any & 2u) && fit_plane_dev(x, y, z, res_atom_idx, lo, hi, bits, 2u, &sc[r])) v |= 2u;
    valid[r] = v;
}

// ---- plane geometry (residues.rs:31-75) ------------------------------------------------------------------------------
__device__ inline double norm3d(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
__device__ inline double fold_deg(double rad) {  // residues.rs:50-53,70-73
    if (rad > 1.57079632679489661923) rad = 3.14159265358979323846 - rad;
    return rad * (180.0 / 3.14159265358979323846264338327950288);
}
__device__ inline double point_dist_d(const PlaneD &p, const double q[3]) { const double v[3] = {q[0] - p.c[0], q[1] - p.c[1], q[2] - p.c[2]}; return norm3d(v); }
__device__ inline double point_angle_d(const PlaneD &p, const double q[3]) {
    const double v[3] = {q[0] - p.c[0], q[1] - p.c[1], q[2] - p.c[2]};
    const double dot = p.n[0] * v[0] + p.n[1] * v[1] + p.n[2] * v[2];
    return fold_deg(acos(dot / (norm3d(p.n) * norm3d(v))));
}
__device__ inline double plane_dihedral_d(const PlaneD &a, const PlaneD &b) {
    const double dot = a.n[0] * b.n[0] + a.n[1] * b.n[1] + a.n[2] * b.n[2];
    return fold_deg(acos(dot / (norm3d(a.n) * norm3d(b.n))));
}

// should_compare_residues (complex.rs:94-131) on prepared keys
struct ResKeyD { int32_t model_serial; uint32_t chain_rank, ord; bool in_l, in_r; };
__device__ inline bool compare_residues_d(const ResKeyD &a, const ResKeyD &b, bool symmetric) {
    if (a.model_serial != b.model_serial) return false;
    if (!((a.in_l && b.in_r) || (b.in_l && a.in_r))) return false;
    if (a.chain_rank == b.chain_rank) {
        if (symmetric) return (b.ord > 1u) && (a.ord < b.ord - 1u);
        const bool neigh = (a.ord == 0u) ? (b.ord == a.ord || b.ord == a.ord + 1u) : (b.ord == a.ord - 1u || b.ord == a.ord || b.ord == a.ord + 1u);
        return !neigh;
    }
    return !(symmetric && a.in_r && b.in_r && a.in_l && b.in_l && a.chain_rank > b.chain_rank);
}

// ---- rows ------------------------------------------------------------------------------------------------------------
// A row = {from entity, to entity, (f32) distance, interaction code}: the layout of arp_pair.  Entity = atom index, or n + ring index.
__device__ inline void append_row(uint4 *rows, uint32_t *n_rows, uint32_t cap, uint32_t from, uint32_t to, double dist, uint32_t code) {
    const uint32_t p = atomicAdd(n_rows, 1u);
    if (p < cap) rows[p] = make_uint4(from, to, __float_as_uint((float)dist), code);  // mod.rs:148: distance narrowed to f32 at table build
}

// get_ring_atom_contacts (complex.rs:301-352) + find_cation_pi (aromatic.rs:14-29) on the pair pass's cell list.  The reference walks an
// R*-tree of all atoms around every ring centre; here one wave per ring visits the cells its search sphere touches -- a handful of
// x-contiguous slot runs of the cell-sorted records the engine left in its workspace (Fat: coordinates, attribute word, residue ordinal,
// chain, original index) -- and only positively ionizable atoms by RESIDUE name (aromatic.rs:18) go on to the plane arithmetic.
// Two chores ride along (each was a launch of its own, 4-5 us on a table that takes 200): every ring's packed {centre, model serial | flags}
// record for k_ring_ring's sweep, and -- so that ONE 16-byte read-back brings the host both row counts -- the atom-row total of the offset
// scan that ran before this kernel goes next to the ring-row counter (counters[3]).
struct RingPoint { double c[3]; int32_t model_serial; uint32_t flags; };
__global__ __launch_bounds__(64) void k_ring_atom(uint32_t n_rings, const RingEnt *rings, const PlaneD *ring_planes, uint32_t n_atoms, const int32_t *model_serial_of,
                                                  const GridParams *gp, const uint32_t *cell_start, const Fat *fat, double radius, uint4 *rows, uint32_t *n_rows,
                                                  uint32_t cap, RingPoint *pts, const uint32_t *atom_rows_total) {
    const uint32_t e = blockIdx.x;
    if (e >= n_rings) return;
    const RingEnt ring = rings[e];
    const PlaneD pl = ring_planes[ring.src_res];
    if (threadIdx.x == 0u) {
        pts[e] = RingPoint{{pl.c[0], pl.c[1], pl.c[2]}, ring.model_serial, ring.flags};
        if (e == 0u && atom_rows_total) n_rows[3] = *atom_rows_total;
    }
    if (!(ring.flags & 4u)) return;
    const ResKeyD rk{ring.model_serial, ring.chain_rank, ring.ord, (ring.flags & 1u) != 0u, (ring.flags & 2u) != 0u};
    const GridParams g = *gp;
    if (g.model_org || g.n_heavy == 0u) return;  // (a packed batch's grid never reaches the table path)
    const double r2 = radius * radius;  // (the reference only ever uses the square: complex.rs:303)
    const double inv[3] = {g.inv_edge_x, g.inv_edge, g.inv_edge};  // cells are kx times finer along x
    // cells the sphere can touch, per axis: floor(f - rho) .. floor(f + rho) of the centre's cell coordinate f, clamped to the grid
    uint32_t lo[3], hi[3];
    const uint32_t dim[3] = {g.nx, g.ny, g.nz};
    const double org[3] = {g.ox, g.oy, g.oz};
    for (int k = 0; k < 3; k++) {
        const double f = (pl.c[k] - org[k]) * inv[k], rho = fabs(radius) * inv[k] * (1.0 + 1e-9) + 1e-9;
        const double a = floor(f - rho), b = floor(f + rho);
        if (!(b >= 0.0) || !(a <= (double)(dim[k] - 1u))) return;  // the sphere misses the grid (or a non-finite centre)
        lo[k] = a > 0.0 ? (uint32_t)a : 0u;
        hi[k] = b < (double)(dim[k] - 1u) ? (uint32_t)b : dim[k] - 1u;
    }
    const uint32_t n_models = g.nzt / (g.nz + 1u);
    for (uint32_t m = 0; m < n_models; m++) {
        if (model_serial_of[m] != ring.model_serial) continue;  // same serial = same model for the reference (complex.rs:96-98)
        for (uint32_t cz = lo[2]; cz <= hi[2]; cz++)
            for (uint32_t cy = lo[1]; cy <= hi[1]; cy++) {
                const uint32_t row = grid_row(cy, m * (g.nz + 1u) + cz, g.ny, g.nzt, g.sy_shift) * g.nx;
                const uint32_t s0 = cell_start[row + lo[0]], s1 = cell_start[row + hi[0] + 1u];
                for (uint32_t p = s0 + threadIdx.x; p < s1; p += 64u) {
                    const Fat f = fat[p];
                    if (!(f.attr & ARP_ATTR_POS_RESN)) continue;
                    const double q[3] = {f.x, f.y, f.z};
                    const double dx = q[0] - pl.c[0], dy = q[1] - pl.c[1], dz = q[2] - pl.c[2];
                    if (!(dx * dx + dy * dy + dz * dz <= r2)) continue;  // rstar: inclusive (complex.rs:310)
                    const ResKeyD yk{ring.model_serial, f.crm, f.res_ord, (f.attr & ARP_ATTR_LIGAND) != 0u, (f.attr & ARP_ATTR_RECEPTOR) != 0u};
                    if (!compare_residues_d(rk, yk, false)) continue;
                    const double dist = point_dist_d(pl, q), theta = point_angle_d(pl, q);
                    if (theta <= 30.0 && dist <= 4.5) append_row(rows, n_rows, cap, n_atoms + e, f.orig, dist, ARP_CationPi);
                }
            }
    }
}

// get_ring_ring_contacts (complex.rs:354-405) + find_pi_pi (aromatic.rs:33-64): ordered ring pairs, k1 in the ligand set, k2 in
// the receptor set.  The sweep over the other rings reads a packed {centre, model serial | flags} record per ring (32 bytes, coalesced);
// the plane and the ring entry are gathered only for the few rings within 6 A.
__global__ __launch_bounds__(256) void k_ring_ring(uint32_t n_rings, const RingEnt *rings, const PlaneD *ring_planes, const RingPoint *pts, uint32_t n_atoms, uint4 *rows,
                                                   uint32_t *n_rows, uint32_t cap) {
    const uint32_t e1 = blockIdx.x;
    if (e1 >= n_rings) return;
    const RingEnt k1 = rings[e1];
    if (!(k1.flags & 4u) || !(k1.flags & 1u)) return;
    const PlaneD p1 = ring_planes[k1.src_res];
    const ResKeyD r1{k1.model_serial, k1.chain_rank, k1.ord, (k1.flags & 1u) != 0u, (k1.flags & 2u) != 0u};
    for (uint32_t e2 = threadIdx.x; e2 < n_rings; e2 += blockDim.x) {
        const RingPoint q = pts[e2];
        if (!(q.flags & 4u) || !(q.flags & 2u) || q.model_serial != k1.model_serial) continue;
        const double v[3] = {p1.c[0] - q.c[0], p1.c[1] - q.c[1], p1.c[2] - q.c[2]};
        const double dist = norm3d(v);
        if (!(dist <= 6.0)) continue;
        const RingEnt k2 = rings[e2];
        const PlaneD p2 = ring_planes[k2.src_res];
        const ResKeyD r2k{k2.model_serial, k2.chain_rank, k2.ord, (k2.flags & 1u) != 0u, (k2.flags & 2u) != 0u};
        if (!compare_residues_d(r1, r2k, true)) continue;
        const double theta = point_angle_d(p1, p2.c), dih = plane_dihedral_d(p1, p2);
        int code = -1;
        if (dih <= 30.0) { if (theta <= 30.0) code = ARP_PiSandwichStacking; else if (theta <= 60.0) code = ARP_PiDisplacedStacking; else if (theta <= 90.0) code = ARP_PiParallelInPlaneStacking; }
        else if (dih <= 60.0) code = ARP_PiTiltedStacking;
        else if (dih <= 90.0) { if (theta >= 30.0 && theta < 60.0) code = ARP_PiLStacking; else if (dist <= 5.0) code = ARP_PiTStacking; }
        if (code >= 0) append_row(rows, n_rows, cap, n_atoms + e1, n_atoms + e2, dist, (uint32_t)code);
    }
}

// atom-atom rows: one per set bit of the pair's kind word (complex.rs:217-296), in pair order
// (launched over n_pairs + 1 items: the extra one is the scan's closing zero; the first 64 threads clear the call's counters -- two memsets less)
__global__ __launch_bounds__(256) void k_count_bits(const arp_pair *pairs, uint32_t n_pairs, uint32_t *bits, uint32_t *counters) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < 64u) counters[p] = 0u;
    if (p < n_pairs) bits[p] = (uint32_t)__popc(pairs[p].kind);
    else if (p == n_pairs) bits[p] = 0u;
}
// (items n_pairs .. n_pairs + n_ring_rows - 1 move the ring kernels' rows behind the atom rows)
__global__ __launch_bounds__(256) void k_expand_rows(const arp_pair *pairs, uint32_t n_pairs, const uint32_t *first, uint4 *rows, uint32_t cap, const uint4 *ring_rows,
                                                     uint32_t n_ring_rows, uint32_t n_atom_rows) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) {
        if (p - n_pairs < n_ring_rows && n_atom_rows + (p - n_pairs) < cap) rows[n_atom_rows + (p - n_pairs)] = ring_rows[p - n_pairs];
        return;
    }
    const arp_pair q = pairs[p];
    uint32_t o = first[p];
    for (uint32_t b = q.kind; b; b &= b - 1u, ++o)
        if (o < cap) rows[o] = make_uint4(q.i, q.j, __float_as_uint(q.dist), (uint32_t)(__ffs((int)b) - 1));
}

// ---- entity ranks and the 10-key sort (mod.rs:120-134) ----------------------------------------------------------------
// model, from_chain, to_chain, from_resi, from_altloc, from_atomi, to_resi, to_altloc, to_atomi, interaction; ties (the reference's
// sort is unstable there) by from_insertion, to_insertion, distance.  (resi, altloc, atomi) of an entity collapse into ONE dense rank
// INSIDE ITS CHAIN, computed by sorting the entities once per structure: rows are only ever compared on these three keys when their
// chains are equal, and a rank inside the chain needs bits(largest chain) instead of bits(all entities) -- which is what lets all ten keys
// of a row share one 64-bit radix key (a million entities in 1500 chains: 48 bits instead of 68).
__device__ inline uint32_t bias(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ inline uint32_t ent_chain(uint32_t e, uint32_t n_atoms, const uint32_t *chain_rank, const RingEnt *rings) {
    return e < n_atoms ? (uint32_t)chain_rank[e] : rings[e - n_atoms].chain_rank;
}
__global__ __launch_bounds__(256) void k_ent_key(uint32_t n_ent, const EntKey *atom_keys, uint32_t n_atoms, const EntKey *ring_keys, const uint32_t *chain_rank,
                                                 const RingEnt *rings, int pass, const uint32_t *ids, unsigned long long *key) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_ent) return;
    const uint32_t e = ids ? ids[p] : p;
    if (pass == 2) { key[p] = ent_chain(e, n_atoms, chain_rank, rings); return; }
    const EntKey k = e < n_atoms ? atom_keys[e] : ring_keys[e - n_atoms];
    key[p] = pass == 0 ? (unsigned long long)bias(k.atomi) : (((unsigned long long)bias(k.resi) << 32) | k.altloc);
}
// entities sorted by (chain, resi, altloc, atomi): flag = the key differs from the predecessor's
__global__ __launch_bounds__(256) void k_ent_flags(uint32_t n_ent, const EntKey *atom_keys, uint32_t n_atoms, const EntKey *ring_keys, const uint32_t *chain_rank,
                                                   const RingEnt *rings, const uint32_t *ids, uint32_t *flag) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_ent) return;
    auto key_of = [&](uint32_t e) { return e < n_atoms ? atom_keys[e] : ring_keys[e - n_atoms]; };
    uint32_t f = 0;
    if (p > 0) {
        const EntKey a = key_of(ids[p - 1]), b = key_of(ids[p]);
        f = (a.resi != b.resi || a.altloc != b.altloc || a.atomi != b.atomi || ent_chain(ids[p - 1], n_atoms, chain_rank, rings) != ent_chain(ids[p], n_atoms, chain_rank, rings)) ? 1u : 0u;
    }
    flag[p] = f;
}
// scan = dense rank over all entities; base[chain] = the rank its first entity got
__global__ __launch_bounds__(256) void k_ent_base(uint32_t n_ent, uint32_t n_atoms, const uint32_t *chain_rank, const RingEnt *rings, const uint32_t *ids, const uint32_t *scan,
                                                  uint32_t *base) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_ent) return;
    const uint32_t c = ent_chain(ids[p], n_atoms, chain_rank, rings);
    if (p == 0 || ent_chain(ids[p - 1], n_atoms, chain_rank, rings) != c) base[c] = scan[p];
}
__global__ __launch_bounds__(256) void k_ent_rank(uint32_t n_ent, uint32_t n_atoms, const uint32_t *chain_rank, const RingEnt *rings, const uint32_t *ids, const uint32_t *scan,
                                                  const uint32_t *base, uint32_t *rank, uint32_t *max_rank) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = 0;
    if (p < n_ent) {
        const uint32_t e = ids[p];
        r = scan[p] - base[ent_chain(e, n_atoms, chain_rank, rings)];
        rank[e] = r;
    }
    for (int off = 32; off; off >>= 1) r = max(r, (uint32_t)__shfl_xor((int)r, off));
    if ((threadIdx.x & 63u) == 0u && r) atomicMax(max_rank, r);
}

__global__ __launch_bounds__(256) void k_iota(uint32_t n_items, uint32_t *v) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_items) v[p] = p;
}

struct SortTables {
    uint8_t name_rank[32];   // interaction code -> rank of its name (the column is a string in the reference)
    uint32_t rank_bits, chain_bits;   // widths of an entity rank / a chain rank in the merged keys
};
__global__ __launch_bounds__(256) void k_row_key(uint32_t n_rows, const uint4 *rows, const uint32_t *perm, int pass, uint32_t n_atoms, const EntKey *atom_keys,
                                                 const EntKey *ring_keys, const uint32_t *ent_rank, const uint32_t *chain_rank, const uint32_t *model, const uint32_t *model_rank,
                                                 const RingEnt *rings, SortTables tb, unsigned long long *key, uint32_t *perm_init) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_rows) return;
    if (perm_init) perm_init[p] = p;  // the first pass starts from the identity: written here (perm == nullptr), not by a launch of its own
    const uint4 r = rows[perm ? perm[p] : p];
    auto ins_of = [&](uint32_t e) { return e < n_atoms ? atom_keys[e].icode : ring_keys[e - n_atoms].icode; };
    auto chain_of = [&](uint32_t e) { return e < n_atoms ? (uint32_t)chain_rank[e] : rings[e - n_atoms].chain_rank; };
    unsigned long long k = 0;
    switch (pass) {
        case 0: k = r.z; break;                                                                           // distance (non-negative f32: bit order == value order)
        case 1: k = ((unsigned long long)ins_of(r.x) << 32) | ins_of(r.y); break;                         // from_insertion, to_insertion
        case 2: k = ((unsigned long long)ent_rank[r.y] << 5) | tb.name_rank[r.w & 31u]; break;            // to_resi, to_altloc, to_atomi, interaction
        case 3: k = ent_rank[r.x]; break;                                                                 // from_resi, from_altloc, from_atomi
        default: {                                                                                        // model, from_chain, to_chain (pass 4) ...
            const uint32_t mr = r.x < n_atoms ? model_rank[model[r.x]] : rings[r.x - n_atoms].model_rank;
            k = ((((unsigned long long)mr << tb.chain_bits) | chain_of(r.x)) << tb.chain_bits) | chain_of(r.y);
            if (pass >= 5) k = (k << tb.rank_bits) | ent_rank[r.x];                                       // ... + the keys of pass 3 (pass 5)
            if (pass == 6) k = (k << (tb.rank_bits + 5u)) | ((unsigned long long)ent_rank[r.y] << 5) | tb.name_rank[r.w & 31u];  // ... + those of pass 2 (pass 6)
        }
    }
    key[p] = k;
}

// The ten keys tie only between rows that mention the same two entities with the same interaction: twice the same ring-ring pair of a
// residue with two rings, or atoms that share (resi, altloc, atomi).  Instead of sorting EVERY row by insertion codes and distance first
// (five more radix passes over the table), the sorted rows are inspected once: a row whose neighbour carries the same keys finds its run
// and takes the place its (from_insertion, to_insertion, distance, position) earns inside it.  Runs longer than kTieRun raise `overflow`
// and the host sorts again the long way.
constexpr uint32_t kTieRun = 64;
__global__ __launch_bounds__(256) void k_tie_fix(uint32_t n_rows, const uint4 *rows, const uint32_t *perm, const unsigned long long *sorted_key, uint32_t n_atoms,
                                                 const EntKey *atom_keys, const EntKey *ring_keys, const uint32_t *ent_rank, const uint32_t *chain_rank, const uint32_t *model,
                                                 const uint32_t *model_rank, const RingEnt *rings, uint32_t *perm_out, uint32_t *overflow) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_rows) return;
    auto same = [&](uint32_t a, uint32_t b) {  // rows at sorted positions a, b carry the same ten keys
        if (sorted_key) return sorted_key[a] == sorted_key[b];
        const uint4 ra = rows[perm[a]], rb = rows[perm[b]];
        if (ra.w != rb.w || ent_rank[ra.x] != ent_rank[rb.x] || ent_rank[ra.y] != ent_rank[rb.y]) return false;
        if (ent_chain(ra.x, n_atoms, chain_rank, rings) != ent_chain(rb.x, n_atoms, chain_rank, rings) ||
            ent_chain(ra.y, n_atoms, chain_rank, rings) != ent_chain(rb.y, n_atoms, chain_rank, rings)) return false;
        const uint32_t ma = ra.x < n_atoms ? model_rank[model[ra.x]] : rings[ra.x - n_atoms].model_rank, mb = rb.x < n_atoms ? model_rank[model[rb.x]] : rings[rb.x - n_atoms].model_rank;
        return ma == mb;
    };
    const bool tie_prev = p > 0 && same(p - 1, p), tie_next = p + 1 < n_rows && same(p, p + 1);
    if (!tie_prev && !tie_next) { perm_out[p] = perm[p]; return; }
    uint32_t lo = p, hi = p + 1;  // the run [lo, hi)
    while (lo > 0 && p - lo < kTieRun && same(lo - 1, lo)) lo--;
    while (hi < n_rows && hi - p < kTieRun && same(hi - 1, hi)) hi++;
    if (p - lo >= kTieRun || hi - p >= kTieRun) { *overflow = 1u; perm_out[p] = perm[p]; return; }
    auto ins_of = [&](uint32_t e) { return e < n_atoms ? atom_keys[e].icode : ring_keys[e - n_atoms].icode; };
    auto tie_key = [&](uint32_t q, unsigned long long *ins, uint32_t *dist) {
        const uint4 r = rows[perm[q]];
        *ins = ((unsigned long long)ins_of(r.x) << 32) | ins_of(r.y); *dist = r.z;  // (non-negative f32: bit order == value order)
    };
    unsigned long long my_ins; uint32_t my_dist;
    tie_key(p, &my_ins, &my_dist);
    uint32_t before = 0;
    for (uint32_t q = lo; q < hi; q++) {
        if (q == p) continue;
        unsigned long long ins; uint32_t dist;
        tie_key(q, &ins, &dist);
        const bool less = ins != my_ins ? ins < my_ins : (dist != my_dist ? dist < my_dist : q < p);
        before += less ? 1u : 0u;
    }
    perm_out[lo + before] = perm[p];
}

// final order + collect_sc_stats (complex.rs:137-174): res1 = ligand residue, res2 = receptor residue
__global__ __launch_bounds__(256) void k_finish_rows(uint32_t n_rows, const uint4 *rows, const uint32_t *perm, uint32_t n_atoms, const uint32_t *atom_sc_src, const RingEnt *rings,
                                                     const PlaneD *sc_planes, const uint8_t *valid, uint4 *out_rows, float4 *out_sc, const uint32_t *tie_overflow,
                                                     uint32_t *flag_out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_rows) return;
    if (p == 0u) *flag_out = *tie_overflow;  // (k_tie_fix ran before: the flag travels with the rows, in the one copy that fetches them)
    const uint4 r = rows[perm[p]];
    out_rows[p] = r;
    const uint32_t s1 = r.x < n_atoms ? atom_sc_src[r.x] : rings[r.x - n_atoms].sc_src, s2 = r.y < n_atoms ? atom_sc_src[r.y] : rings[r.y - n_atoms].sc_src;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s1 != ARP_NONE && s2 != ARP_NONE && (valid[s1] & 2u) && (valid[s2] & 2u)) {
        const PlaneD p1 = sc_planes[s1], p2 = sc_planes[s2];
        sc = make_float4((float)point_dist_d(p1, p2.c), (float)plane_dihedral_d(p1, p2), (float)point_angle_d(p1, p2.c), 1.0f);
    }
    out_sc[p] = sc;
}

// ---- PDB-sized tables in ONE launch (round 5) -----------------------------------------------------------------------------------------
// A table of a few thousand rows is a chain of launches that each last 4-5 us whatever they do: bit count, offset scan (2), row expansion, key,
// rocPRIM's block sort + three merges, tie fix, finish, the copy of the rows -- 54 of 6bft's 134 us were the sort alone.  Up to kSmallRows rows one
// workgroup does all of it (the launcher takes this path up to kSmallPairs pairs, see device_table): the rows of a pair per set bit of its kind word (complex.rs:217-296) at the offsets of a block-wide scan, the ring rows
// behind them, the ten sort keys of mod.rs:120-134 merged into one word (as k_row_key's pass 6) with the row index below them -- so that a plain
// compare-exchange network sorts (keys, position), which is what the stable radix passes of the general path produce --, a bitonic sort of those
// words in LDS, the tie pass on neighbours (k_tie_fix), the side-chain plane statistics (k_finish_rows), and the finished table written STRAIGHT
// INTO THE PINNED LANDING BUFFER the host reads after the stream has drained (no copy command).  What does not fit (more rows, keys wider than
// 64 - kSmallIdxBits bits, a tie run longer than kTieRun) is reported in the header and the general path runs instead.
constexpr uint32_t kSmallRows = 4096, kSmallThreads = 1024, kSmallIdxBits = 12, kSmallPairs = 2048;
struct SmallHeader { uint32_t n_rows, status, n_atom_rows, n_ring_rows; };  // status: 0 = done; 1 = too many rows; 2 = a tie run too long; 3 = more ring rows than reserved

// Bitonic sort (ascending) of n_pad = E x (active threads) 64-bit words, thread t holding the E consecutive words t E .. t E + E - 1 in registers.
// A compare-exchange step at distance j pairs word g with word g ^ j: inside the thread's own registers for j < E, with the same register of
// thread t ^ (j / E) beyond -- by a wave shuffle while that thread is in the same wave, through LDS (two barriers) only for the steps whose
// partner sits in another wave: 6 of the 78 steps of 4096 words.  (The first version of k_table_small ran all 91 through LDS, one
// compare-exchange at a time per thread: 68 us on 6bft's 7236 rows.)
template <uint32_t E>
__device__ __forceinline__ void bitonic_sort_block(unsigned long long (&v)[E], unsigned long long *buf, uint32_t n_pad, uint32_t tid) {
    const uint32_t base = tid * E;
    const bool active = base < n_pad;
    for (uint32_t k = 2; k <= n_pad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                unsigned long long other[E];
                const uint32_t pj = j / E;
                if (pj < 64u) {
#pragma unroll
                    for (uint32_t e = 0; e < E; e++) other[e] = __shfl_xor(v[e], (int)pj);
                } else {
                    __syncthreads();  // (the previous exchange's reads are done)
                    if (active) {
#pragma unroll
                        for (uint32_t e = 0; e < E; e++) buf[base + e] = v[e];
                    }
                    __syncthreads();
#pragma unroll
                    for (uint32_t e = 0; e < E; e++) other[e] = active ? buf[(base + e) ^ j] : v[e];
                }
                // (j >= E and k > j: bits j and k of word t E + e are those of t E -- one decision per thread and step)
                const bool keep_min = ((base & k) == 0u) == ((base & j) == 0u);  // ascending stretch and the lower partner, or descending and the upper one
#pragma unroll
                for (uint32_t e = 0; e < E; e++) {
                    const unsigned long long a = v[e], b = other[e];
                    v[e] = ((a < b) == keep_min) ? a : b;
                }
            } else {
                // the steps inside the thread's own words: all of them at once, with compile-time distances (a register array indexed by the
                // loop variable j would live in scratch memory: the first blocked version of this sort took 350 us that way)
#pragma unroll
                for (uint32_t jj = E / 2u; jj > 0u; jj >>= 1) {
                    if (jj > j) continue;  // (this merge starts below jj)
#pragma unroll
                    for (uint32_t e = 0; e < E; e++) {
                        if (e & jj) continue;
                        const bool up = ((base + e) & k) == 0u;
                        const unsigned long long a = v[e], b = v[e | jj];
                        const bool swap = (a > b) == up;
                        v[e] = swap ? b : a; v[e | jj] = swap ? a : b;
                    }
                }
                break;  // (j < E: the rest of this merge was done above)
            }
        }
    }
}

__global__ __launch_bounds__(kSmallThreads) void k_table_small(const arp_pair *pairs, uint32_t n_pairs, const uint4 *ring_rows, const uint32_t *ring_row_count, uint32_t ring_rows_cap,
                                                               uint4 *rows, uint32_t n_atoms, const EntKey *atom_keys, const EntKey *ring_keys, const uint32_t *ent_rank,
                                                               const uint32_t *chain_rank, const uint32_t *model, const uint32_t *model_rank, const RingEnt *rings, SortTables tb,
                                                               const uint32_t *atom_sc_src, const PlaneD *sc_planes, const uint8_t *valid, char *out, unsigned long long *stamps) {
    __shared__ unsigned long long skey[kSmallRows];     // the sort words (64 KB): exchange buffer of the sort, then the sorted table
    __shared__ uint16_t sperm[kSmallRows];              // final order: row index per table position
    __shared__ uint32_t wave_sum[kSmallThreads / 64];
    __shared__ uint32_t s_flag;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    SmallHeader *head = reinterpret_cast<SmallHeader *>(out);
    const uint32_t n_ring_rows = ring_row_count ? *ring_row_count : 0u;
    if (tid == 0u) s_flag = 0u;
    auto stamp = [&](int k) { if (stamps && tid == 0u) stamps[k] = wall_clock64(); };  // (arp_debug_set("timing", 1): where the launch's time goes, 100 MHz ticks)
    stamp(0);
    // 1. rows per pair, their offsets (thread t: pairs t kPer .. in order, so the offsets are those of the general path's scan); all of a thread's
    // pairs are requested before the first is looked at -- one workgroup has nothing else to hide a round trip to memory behind
    constexpr uint32_t kPer = kSmallRows / kSmallThreads;
    uint4 pr[kPer];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t u = 0; u < kPer; u++) {
        const uint32_t p = tid * kPer + u;
        pr[u] = reinterpret_cast<const uint4 *>(pairs)[p < n_pairs ? p : 0u];
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; u++) {
        if (tid * kPer + u >= n_pairs) pr[u].w = 0u;
        mine += (uint32_t)__popc(pr[u].w);
    }
    uint32_t inc = mine;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off); if (lane >= (uint32_t)off) inc += t; }
    if (lane == 63u) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, n_atom_rows = 0;
    for (uint32_t k = 0; k < kSmallThreads / 64u; k++) { const uint32_t t = wave_sum[k]; if (k < wave) before += t; n_atom_rows += t; }
    const uint32_t n_rows = n_atom_rows + n_ring_rows;
    if (n_rows > kSmallRows || n_ring_rows > ring_rows_cap) {
        if (tid == 0u) *head = SmallHeader{n_rows, n_ring_rows > ring_rows_cap ? 3u : 1u, n_atom_rows, n_ring_rows};
        return;
    }
    // 2. the rows (global scratch: 128 KB would not fit LDS next to the sort words), atom rows first, ring rows behind them
    {
        uint32_t o = before + inc - mine;
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++)
            for (uint32_t b = pr[u].w; b; b &= b - 1u, ++o) rows[o] = make_uint4(pr[u].x, pr[u].y, pr[u].z, (uint32_t)(__ffs((int)b) - 1));
        for (uint32_t k = tid; k < n_ring_rows; k += kSmallThreads) rows[n_atom_rows + k] = ring_rows[k];
    }
    __syncthreads();  // (workgroup-scope: the rows written above are read by other waves of this block below)
    stamp(1);
    // 3. one sort word per row: the ten keys (k_row_key pass 6) above the row's index; padding sorts last.  Thread t holds the words of rows
    // t E .. t E + E - 1 (E = n_pad / 1024, at least 1); the gathers of a thread's rows are in flight together.
    auto chain_of = [&](uint32_t e) { return e < n_atoms ? (uint32_t)chain_rank[e] : rings[e - n_atoms].chain_rank; };
    uint32_t n_pad = 64u;
    while (n_pad < n_rows) n_pad <<= 1;
    const uint32_t E = n_pad > kSmallThreads ? n_pad / kSmallThreads : 1u;
    unsigned long long v[kPer];
    {
        uint4 r[kPer];
#pragma unroll
        for (uint32_t e = 0; e < kPer; e++) { const uint32_t p = tid * E + e; r[e] = rows[(e < E && p < n_rows) ? p : 0u]; }
#pragma unroll
        for (uint32_t e = 0; e < kPer; e++) {
            const uint32_t p = tid * E + e;
            v[e] = ~0ull;
            if (e < E && p < n_rows) {
                const uint32_t mr = r[e].x < n_atoms ? model_rank[model[r[e].x]] : rings[r[e].x - n_atoms].model_rank;
                unsigned long long k = ((((unsigned long long)mr << tb.chain_bits) | chain_of(r[e].x)) << tb.chain_bits) | chain_of(r[e].y);
                k = (k << tb.rank_bits) | ent_rank[r[e].x];
                k = (k << (tb.rank_bits + 5u)) | ((unsigned long long)ent_rank[r[e].y] << 5) | tb.name_rank[r[e].w & 31u];
                v[e] = (k << kSmallIdxBits) | p;
            }
        }
    }
    stamp(2);
    // 4. the sort
    switch (E) {
        case 1: { unsigned long long w1[1] = {v[0]}; bitonic_sort_block<1>(w1, skey, n_pad, tid); v[0] = w1[0]; break; }
        case 2: { unsigned long long w2[2] = {v[0], v[1]}; bitonic_sort_block<2>(w2, skey, n_pad, tid); v[0] = w2[0]; v[1] = w2[1]; break; }
        default: static_assert(kPer == 4, "one instantiation per words-per-thread count"); bitonic_sort_block<kPer>(v, skey, n_pad, tid);
    }
    __syncthreads();  // (the sort's last exchange is done with skey)
#pragma unroll
    for (uint32_t e = 0; e < kPer; e++)
        if (e < E && tid * E + e < n_pad) skey[tid * E + e] = v[e];
    __syncthreads();
    stamp(3);
    // 5. rows whose ten keys tie take the place their (from_insertion, to_insertion, distance, position) earns inside their run (k_tie_fix)
    auto ins_of = [&](uint32_t e) { return e < n_atoms ? atom_keys[e].icode : ring_keys[e - n_atoms].icode; };
    constexpr unsigned long long kIdxMask = (1ull << kSmallIdxBits) - 1ull;
    for (uint32_t p = tid; p < n_rows; p += kSmallThreads) {
        auto same = [&](uint32_t a, uint32_t b) { return (skey[a] >> kSmallIdxBits) == (skey[b] >> kSmallIdxBits); };
        const uint32_t me = (uint32_t)(skey[p] & kIdxMask);
        const bool tie_prev = p > 0u && same(p - 1u, p), tie_next = p + 1u < n_rows && same(p, p + 1u);
        if (!tie_prev && !tie_next) { sperm[p] = (uint16_t)me; continue; }
        uint32_t lo = p, hi = p + 1u;
        while (lo > 0u && p - lo < kTieRun && same(lo - 1u, lo)) lo--;
        while (hi < n_rows && hi - p < kTieRun && same(hi - 1u, hi)) hi++;
        if (p - lo >= kTieRun || hi - p >= kTieRun) { s_flag = 1u; continue; }
        auto tie_key = [&](uint32_t q, unsigned long long *ins, uint32_t *dist) {
            const uint4 r = rows[(uint32_t)(skey[q] & kIdxMask)];
            *ins = ((unsigned long long)ins_of(r.x) << 32) | ins_of(r.y); *dist = r.z;
        };
        unsigned long long my_ins; uint32_t my_dist;
        tie_key(p, &my_ins, &my_dist);
        uint32_t ahead = 0;
        for (uint32_t q = lo; q < hi; q++) {
            if (q == p) continue;
            unsigned long long ins; uint32_t dist;
            tie_key(q, &ins, &dist);
            const bool less = ins != my_ins ? ins < my_ins : (dist != my_dist ? dist < my_dist : q < p);
            ahead += less ? 1u : 0u;
        }
        sperm[lo + ahead] = (uint16_t)me;
    }
    __syncthreads();
    if (s_flag) {
        if (tid == 0u) *head = SmallHeader{n_rows, 2u, n_atom_rows, n_ring_rows};
        return;
    }
    stamp(4);
    // 6. final order + the side-chain plane statistics (k_finish_rows), straight into the host's landing buffer; again every load of a thread's
    // rows is asked for before the first is used
    uint4 *out_rows = reinterpret_cast<uint4 *>(out + 16);
    float4 *out_sc = reinterpret_cast<float4 *>(out + 16 + (size_t)n_rows * 16u);
    {
        uint4 r[kPer];
        uint32_t s1[kPer], s2[kPer];
        uint8_t ok[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) { const uint32_t p = tid + u * kSmallThreads; r[u] = rows[p < n_rows ? sperm[p] : 0u]; }
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) {
            s1[u] = r[u].x < n_atoms ? atom_sc_src[r[u].x] : rings[r[u].x - n_atoms].sc_src;
            s2[u] = r[u].y < n_atoms ? atom_sc_src[r[u].y] : rings[r[u].y - n_atoms].sc_src;
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) ok[u] = (s1[u] != ARP_NONE && s2[u] != ARP_NONE) ? (uint8_t)(valid[s1[u]] & valid[s2[u]] & 2u) : (uint8_t)0;
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) {
            const uint32_t p = tid + u * kSmallThreads;
            if (p >= n_rows) continue;
            out_rows[p] = r[u];
            float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {
                const PlaneD p1 = sc_planes[s1[u]], p2 = sc_planes[s2[u]];
                sc = make_float4((float)point_dist_d(p1, p2.c), (float)plane_dihedral_d(p1, p2), (float)point_angle_d(p1, p2.c), 1.0f);
            }
            out_sc[p] = sc;
        }
    }
    __syncthreads();
    stamp(5);
    if (tid == 0u) *head = SmallHeader{n_rows, 0u, n_atom_rows, n_ring_rows};
}

// ---- host orchestration ----------------------------------------------------------------------------------------------
namespace {
struct Bump {  // carve a scratch block (256-byte aligned pieces)
    char *base; uint64_t off = 0, cap;
    template <class T> T *take(uint64_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255u) & ~255ull;
        return p;
    }
};
uint64_t al(uint64_t bytes) { return (bytes + 255u) & ~255ull; }
}  // namespace

arp_status device_planes(arp_context *ctx, const DevStructure &ds, std::vector<double> *planes, std::vector<uint8_t> *valid) {
    hipStream_t st = (hipStream_t)context_stream(ctx);
    const uint64_t nr = ds.n_res;
    char *dev = nullptr, *pin = nullptr;
    arp_status s = context_scratch(ctx, 0, 2 * al(nr * sizeof(PlaneD)) + al(nr) + 4096, 0, &dev, &pin);
    if (s != ARP_OK) return s;
    Bump b{dev, 0, 0};
    PlaneD *ring = b.take<PlaneD>(nr), *sc = b.take<PlaneD>(nr);
    uint8_t *v = b.take<uint8_t>(nr);
    if (nr) hipLaunchKernelGGL(k_fit_planes, dim3((uint32_t)((nr + 127) / 128)), dim3(128), 0, st, (uint32_t)nr, (const uint32_t *)ds.res_atom_ptr, (const uint32_t *)ds.res_atom_idx,
                               (const uint8_t *)ds.plane_bits, (const double *)ds.x, (const double *)ds.y, (const double *)ds.z, ring, sc, v);
    TRY_HIP(hipGetLastError());
    std::vector<PlaneD> hr(nr), hs(nr);
    valid->assign(nr, 0);
    TRY_HIP(hipMemcpyAsync(hr.data(), ring, nr * sizeof(PlaneD), hipMemcpyDeviceToHost, st));
    TRY_HIP(hipMemcpyAsync(hs.data(), sc, nr * sizeof(PlaneD), hipMemcpyDeviceToHost, st));
    TRY_HIP(hipMemcpyAsync(valid->data(), v, nr, hipMemcpyDeviceToHost, st));
    TRY_HIP(hipStreamSynchronize(st));
    planes->assign(12 * nr, 0.0);
    for (uint64_t r = 0; r < nr; r++) {
        if ((*valid)[r] & 1u) memcpy(&(*planes)[12 * r], &hr[r], sizeof(PlaneD));
        if ((*valid)[r] & 2u) memcpy(&(*planes)[12 * r + 6], &hs[r], sizeof(PlaneD));
    }
    return ARP_OK;
}

arp_status device_table(arp_context *ctx, DevStructure &ds, const std::vector<RingEnt> &rings, const std::vector<EntKey> &ring_keys, const arp_pair *pairs_dev,
                        uint64_t n_pairs, double dist_cutoff, TableRowsHost *out) {
    hipStream_t st = (hipStream_t)context_stream(ctx);
    const bool timing = g_debug.timing != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "    device_table %-20s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const uint64_t n = ds.n, nr = ds.n_res, n_rings = rings.size(), n_ent = n + n_rings;
    if (n_pairs > 0x7FFFFFF0ull / 8 || n_ent > 0x7FFFFFF0ull) { set_error("table too large for 32-bit row indices"); return ARP_ERR_BAD_INPUT; }
    // rows: <= 5 per pair in principle, but a contacts-only pair carries 1.06 rows on average; sized after the bit count below.
    // First stage: everything whose size is known up front.
    size_t cub_scan = 0, cub_sort_ent = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, cub_scan, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)std::max<uint64_t>(n_pairs, n_ent) + 1);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_sort_ent, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)n_ent, 0, 64);
    const uint64_t ring_rows_cap = 64 * n_rings + 1024;  // each ring meets a handful of cations / rings; checked below
    uint64_t need = 2 * al(nr * sizeof(PlaneD)) + al(nr) + al(n_rings * sizeof(RingEnt)) + al(n_rings * sizeof(EntKey)) + al(n_rings * 32) + 4096 + al((n_pairs + 1) * 4) * 2 +
                    al(std::max(cub_scan, cub_sort_ent)) + al(n_ent * 8) * 2 + al(n_ent * 4) * 4 + al(ring_rows_cap * 16);
    char *dev = nullptr, *pin = nullptr;
    const uint64_t pin_bytes = al(n_rings * sizeof(RingEnt)) + al(n_rings * sizeof(EntKey)) + 4096;
    arp_status s = context_scratch(ctx, 0, need, pin_bytes, &dev, &pin);
    if (s != ARP_OK) return s;
    Bump b{dev, 0, need};
    // planes and entity ranks depend on the structure alone: computed by the first call, kept with the resident copy
    const bool derive = !ds.derived || ds.derived_n_ent != n_ent;
    if (derive) {
        if (ds.derived) { (void)hipFree(ds.derived); ds.derived = nullptr; }
        const uint64_t bytes = 2 * al(nr * sizeof(PlaneD)) + al(nr) + al(n_ent * 4) + 1024;
        TRY_HIP(hipMalloc((void **)&ds.derived, bytes));
        Bump db{ds.derived, 0, bytes};
        ds.ring_pl = db.take<PlaneD>(nr); ds.sc_pl = db.take<PlaneD>(nr); ds.pl_valid = db.take<uint8_t>(nr); ds.ent_rank = db.take<uint32_t>(n_ent);
        ds.derived_n_ent = n_ent;
    }
    PlaneD *ring_pl = (PlaneD *)ds.ring_pl, *sc_pl = (PlaneD *)ds.sc_pl;
    uint8_t *valid = ds.pl_valid;
    // the ring entities live on the device with the structure; they are sent again when they change (the chain groups are part of them)
    RingEnt *d_rings = nullptr;
    EntKey *d_ring_keys = nullptr;
    if (n_rings) {
        const uint64_t rb = n_rings * sizeof(RingEnt), kb = n_rings * sizeof(EntKey), tot = al(rb) + kb;
        bool send = ds.rings_host.size() != tot || memcmp(ds.rings_host.data(), rings.data(), rb) != 0 || memcmp(ds.rings_host.data() + al(rb), ring_keys.data(), kb) != 0;
        if (!ds.rings_block || ds.rings_cap < tot) {
            if (ds.rings_block) { (void)hipStreamSynchronize(st); (void)hipFree(ds.rings_block); ds.rings_block = nullptr; }
            TRY_HIP(hipMalloc((void **)&ds.rings_block, tot + tot / 4));
            ds.rings_cap = tot + tot / 4; send = true;
        }
        d_rings = reinterpret_cast<RingEnt *>(ds.rings_block); d_ring_keys = reinterpret_cast<EntKey *>(ds.rings_block + al(rb));
        if (send) {
            ds.rings_host.assign(tot, 0);
            memcpy(ds.rings_host.data(), rings.data(), rb); memcpy(ds.rings_host.data() + al(rb), ring_keys.data(), kb);
            memcpy(pin, ds.rings_host.data(), tot);
            TRY_HIP(hipMemcpyAsync(ds.rings_block, pin, tot, hipMemcpyHostToDevice, st));
        }
    }
    RingPoint *ring_pts = b.take<RingPoint>(n_rings);
    uint32_t *counters = b.take<uint32_t>(64);  // [0] rows, [1] tie runs too long for k_tie_fix, [2] largest entity rank
    uint32_t *bits = b.take<uint32_t>(n_pairs + 1), *first = b.take<uint32_t>(n_pairs + 1);
    char *cub_tmp = b.take<char>(std::max(cub_scan, cub_sort_ent));
    unsigned long long *ek0 = b.take<unsigned long long>(n_ent), *ek1 = b.take<unsigned long long>(n_ent);
    uint32_t *eid0 = b.take<uint32_t>(n_ent), *eid1 = b.take<uint32_t>(n_ent), *eflag = b.take<uint32_t>(n_ent), *ent_rank = ds.ent_rank;
    uint4 *ring_rows = b.take<uint4>(ring_rows_cap);  // the ring kernels' rows, until the row buffers exist (they are sized by BOTH row counts: one read-back)
    volatile uint32_t *counts_host = reinterpret_cast<volatile uint32_t *>(pin + pin_bytes - 64);  // pinned landing of that read-back

    auto grid = [](uint64_t items, uint32_t block) { return dim3((uint32_t)std::max<uint64_t>(1, (items + block - 1) / block)); };
    // f1: plane fits
    if (nr && derive) hipLaunchKernelGGL(k_fit_planes, grid(nr, 128), dim3(128), 0, st, (uint32_t)nr, (const uint32_t *)ds.res_atom_ptr, (const uint32_t *)ds.res_atom_idx,
                               (const uint8_t *)ds.plane_bits, (const double *)ds.x, (const double *)ds.y, (const double *)ds.z, ring_pl, sc_pl, valid);
    // atom-atom rows: bit count -> offsets -> rows (after the total is known)
    // PDB-sized tables (at most kSmallRows pairs, hence checked rows, and keys that leave kSmallIdxBits bits of the sort word) take the
    // one-launch path below; its kernel counts and scans the bits itself, so only the counters are cleared here
    // (measured on 6bft, 7236 rows: the one workgroup sorts for 68 us -- n log^2 n compare-exchanges on ONE compute unit -- and gathers for 40 more, which is
    // no faster than the dozen chip-wide launches it replaces: 0.32 against 0.29 ms per warm call; on 1ubq, 532 rows, the whole kernel is 19 us and the
    // call 0.13 against 0.15 ms.  So the one-launch path takes tables of up to kSmallPairs pairs; the kernel itself holds kSmallRows rows.)
    bool small_try = n_pairs <= kSmallPairs;
    auto launch_bits = [&]() -> arp_status {
        hipLaunchKernelGGL(k_count_bits, grid(n_pairs + 1, 256), dim3(256), 0, st, pairs_dev, (uint32_t)n_pairs, bits, counters);  // (also clears the counters)
        if (n_pairs) {
            size_t tmp = cub_scan;
            TRY_HIP(hipcub::DeviceScan::ExclusiveSum(cub_tmp, tmp, (const uint32_t *)bits, first, (int)n_pairs + 1, st));
        }
        return ARP_OK;
    };
    if (!small_try) { if ((s = launch_bits()) != ARP_OK) return s; }
    else if (n_rings || derive) TRY_HIP(hipMemsetAsync(counters, 0, 64 * sizeof(uint32_t), st));
    // entity ranks: sort the entities by (chain, resi, altloc, atomi), least significant key first; rank = number of key changes since the
    // chain's first entity
    if (derive) {
        const uint32_t *chain = (const uint32_t *)ds.chain_rank;
        hipLaunchKernelGGL(k_iota, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, eid0);
        hipLaunchKernelGGL(k_ent_key, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, (const EntKey *)ds.ent_key, (uint32_t)n, (const EntKey *)d_ring_keys, chain,
                           (const RingEnt *)d_rings, 0, (const uint32_t *)nullptr, ek0);
        size_t tmp = cub_sort_ent;
        TRY_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, tmp, (const unsigned long long *)ek0, ek1, (const uint32_t *)eid0, eid1, (int)n_ent, 0, 32, st));
        hipLaunchKernelGGL(k_ent_key, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, (const EntKey *)ds.ent_key, (uint32_t)n, (const EntKey *)d_ring_keys, chain,
                           (const RingEnt *)d_rings, 1, (const uint32_t *)eid1, ek0);
        tmp = cub_sort_ent;
        TRY_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, tmp, (const unsigned long long *)ek0, ek1, (const uint32_t *)eid1, eid0, (int)n_ent, 0, 64, st));
        hipLaunchKernelGGL(k_ent_key, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, (const EntKey *)ds.ent_key, (uint32_t)n, (const EntKey *)d_ring_keys, chain,
                           (const RingEnt *)d_rings, 2, (const uint32_t *)eid0, ek0);
        tmp = cub_sort_ent;
        {   // (the chain pass: as many key bits as the structure's chain count needs -- 32-bit chain ranks since API v2)
            int chain_key_bits = 1;
            while (chain_key_bits < 32 && (1ull << chain_key_bits) < std::max<uint64_t>(ds.n_chains, 1)) chain_key_bits++;
            TRY_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, tmp, (const unsigned long long *)ek0, ek1, (const uint32_t *)eid0, eid1, (int)n_ent, 0, chain_key_bits, st));
        }
        hipLaunchKernelGGL(k_ent_flags, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, (const EntKey *)ds.ent_key, (uint32_t)n, (const EntKey *)d_ring_keys, chain,
                           (const RingEnt *)d_rings, (const uint32_t *)eid1, eflag);
        tmp = cub_scan;
        TRY_HIP(hipcub::DeviceScan::InclusiveSum(cub_tmp, tmp, (const uint32_t *)eflag, eid0, (int)n_ent, st));
        uint32_t *chain_base = reinterpret_cast<uint32_t *>(ek0);  // (the sort keys are done with: their first n_chains words hold the per-chain bases; n_chains <= n_ent)
        hipLaunchKernelGGL(k_ent_base, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, (uint32_t)n, chain, (const RingEnt *)d_rings, (const uint32_t *)eid1, (const uint32_t *)eid0,
                           chain_base);
        hipLaunchKernelGGL(k_ent_rank, grid(n_ent, 256), dim3(256), 0, st, (uint32_t)n_ent, (uint32_t)n, chain, (const RingEnt *)d_rings, (const uint32_t *)eid1, (const uint32_t *)eid0,
                           (const uint32_t *)chain_base, ent_rank, counters + 2);
        TRY_HIP(hipGetLastError());
        uint32_t max_rank = 0;  // (once per structure: the width of a rank in the rows' sort key)
        TRY_HIP(hipMemcpyAsync(&max_rank, counters + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        TRY_HIP(hipStreamSynchronize(st));
        ds.max_ent_rank = max_rank;
    }
    lap("planes+bits+ranks");
    auto width = [](uint64_t v) { uint32_t b = 1; while (b < 32 && (1ull << b) < v) b++; return b; };  // bits that hold 0 .. v-1
    SortTables tb{};
    {
        int o[ARP_N_INTERACTIONS];
        for (int k = 0; k < ARP_N_INTERACTIONS; k++) o[k] = k;
        std::sort(o, o + ARP_N_INTERACTIONS, [](int a, int c) { return strcmp(arp_interaction_name(a), arp_interaction_name(c)) < 0; });
        for (int k = 0; k < ARP_N_INTERACTIONS; k++) tb.name_rank[o[k]] = (uint8_t)k;
    }
    tb.rank_bits = width((uint64_t)ds.max_ent_rank + 1); tb.chain_bits = width(std::max<uint64_t>(ds.n_chains, 1));
    if (small_try && width(std::max<uint64_t>(ds.n_models, 1)) + 2 * tb.chain_bits + 2 * tb.rank_bits + 5 + kSmallIdxBits > 64) {
        small_try = false;  // (the ten keys + a row index do not fit one word: the general path, from its first kernel)
        if ((s = launch_bits()) != ARP_OK) return s;
    }
    auto launch_rings = [&]() {
        // the cell list of the pair pass that has just run on this context, on these very arrays
        const GridParams *gridp = nullptr; const uint32_t *cell_start = nullptr; const Fat *fat = nullptr;
        if (!context_grid(ctx, ds.x, ds.n, &gridp, &cell_start, &fat)) { set_error("internal error: the context holds no cell list of this structure"); return ARP_ERR_HIP; }
        hipLaunchKernelGGL(k_ring_atom, dim3((uint32_t)n_rings), dim3(64), 0, st, (uint32_t)n_rings, (const RingEnt *)d_rings, (const PlaneD *)ring_pl, (uint32_t)n,
                           (const int32_t *)ds.model_serial_of, gridp, cell_start, fat, dist_cutoff, ring_rows, counters, (uint32_t)ring_rows_cap, ring_pts,
                           n_pairs && !small_try ? (const uint32_t *)(first + n_pairs) : (const uint32_t *)nullptr);
        hipLaunchKernelGGL(k_ring_ring, dim3((uint32_t)n_rings), dim3(256), 0, st, (uint32_t)n_rings, (const RingEnt *)d_rings, (const PlaneD *)ring_pl, (const RingPoint *)ring_pts, (uint32_t)n, ring_rows,
                           counters, (uint32_t)ring_rows_cap);
        return ARP_OK;
    };
    if (small_try) {
        // ONE launch behind the ring kernels, the table written straight into the pinned landing block, one wait for the stream
        const uint64_t land = 16 + (uint64_t)kSmallRows * 32 + 256;
        char *dev2 = nullptr, *pin2 = nullptr;
        if ((s = context_scratch(ctx, 1, al((uint64_t)kSmallRows * 16) + 4096, land, &dev2, &pin2)) != ARP_OK) return s;
        if (n_rings && (s = launch_rings()) != ARP_OK) return s;
        hipLaunchKernelGGL(k_table_small, dim3(1), dim3(kSmallThreads), 0, st, pairs_dev, (uint32_t)n_pairs, (const uint4 *)ring_rows, n_rings ? (const uint32_t *)counters : (const uint32_t *)nullptr,
                           (uint32_t)ring_rows_cap, reinterpret_cast<uint4 *>(dev2), (uint32_t)n, (const EntKey *)ds.ent_key, (const EntKey *)d_ring_keys, (const uint32_t *)ent_rank,
                           (const uint32_t *)ds.chain_rank, (const uint32_t *)ds.model, (const uint32_t *)ds.model_rank, (const RingEnt *)d_rings, tb, (const uint32_t *)ds.atom_sc_src,
                           (const PlaneD *)sc_pl, (const uint8_t *)valid, pin2, timing ? reinterpret_cast<unsigned long long *>(pin2 + land - 128) : (unsigned long long *)nullptr);
        TRY_HIP(hipGetLastError());
        TRY_HIP(hipStreamSynchronize(st));
        SmallHeader head;
        memcpy(&head, pin2, sizeof head);
        lap("one-launch table");
        if (timing) {
            unsigned long long t[6];
            memcpy(t, pin2 + land - 128, sizeof t);
            fprintf(stderr, "      k_table_small (us): rows %.1f  keys %.1f  sort %.1f  ties %.1f  finish %.1f\n", (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01,
                    (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01);
        }
        if (head.status == 0u) {
            out->n = head.n_rows;
            const size_t row_bytes = (size_t)head.n_rows * 16;
            char *heap = (char *)malloc(2 * row_bytes + 64);
            if (!heap) { set_error("out of host memory"); return ARP_ERR_OOM; }
            out->owner = std::shared_ptr<char>(heap, [](char *q) { free(q); });
            memcpy(heap, pin2 + 16, 2 * row_bytes);
            out->rows = reinterpret_cast<TableRow *>(heap);
            out->sc = reinterpret_cast<TableSc *>(heap + row_bytes);
            lap("unpack");
            return ARP_OK;
        }
        if (head.status == 3u) { set_error("internal error: more ring rows than reserved (%u > %llu)", head.n_ring_rows, (unsigned long long)ring_rows_cap); return ARP_ERR_HIP; }
        small_try = false;  // more rows than one workgroup sorts, or a long tie run: the general path, from its first kernel
        if ((s = launch_bits()) != ARP_OK) return s;
    }
    // f1: the ring rows, into a buffer of their own (their number is bounded up front) -- so that the ONE read-back below brings both the
    // atom-row and the ring-row count and the row buffers are sized exactly, without a second wait for the device.
    if (n_rings) {
        if ((s = launch_rings()) != ARP_OK) return s;
    }
    // how many rows?  (one small read-back: the row buffers are sized by it)
    counts_host[0] = counts_host[3] = 0u;
    if (n_rings) TRY_HIP(hipMemcpyAsync((void *)counts_host, counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    else if (n_pairs) TRY_HIP(hipMemcpyAsync((void *)(counts_host + 3), first + n_pairs, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    TRY_HIP(hipStreamSynchronize(st));
    const uint32_t n_ring_rows = counts_host[0], n_atom_rows = counts_host[3];
    if (n_ring_rows > ring_rows_cap) { set_error("internal error: more ring rows than reserved (%u > %llu)", n_ring_rows, (unsigned long long)ring_rows_cap); return ARP_ERR_HIP; }
    const uint32_t n_rows = n_atom_rows + n_ring_rows;
    const uint64_t rows_cap = std::max<uint64_t>(n_rows, 1);
    size_t cub_sort_rows = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_sort_rows, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)rows_cap, 0, 64);
    const uint64_t need2 = al(rows_cap * 16) + al(rows_cap * 32 + 256) + al(rows_cap * 8) * 2 + al(rows_cap * 4) * 2 + al(cub_sort_rows) + 4096;
    char *dev2 = nullptr, *pin2 = nullptr;  // the row-sized buffers live in a scratch slot of their own: the first one must not move
    if ((s = context_scratch(ctx, 1, need2, 2 * al(rows_cap * 16) + 4096, &dev2, &pin2)) != ARP_OK) return s;
    b = Bump{dev2, 0, need2};
    pin = pin2;
    lap("ring rows + scratch");
    // the finished table in ONE device block {rows, sc values, the tie-overflow word}: one copy fetches it
    uint4 *rows = b.take<uint4>(rows_cap);
    char *out_all = b.take<char>(rows_cap * 32 + 256);
    uint4 *out_rows = reinterpret_cast<uint4 *>(out_all);
    float4 *out_sc = reinterpret_cast<float4 *>(out_all + (size_t)n_rows * 16);
    uint32_t *out_flag = reinterpret_cast<uint32_t *>(out_all + (size_t)n_rows * 32);
    unsigned long long *rk0 = b.take<unsigned long long>(rows_cap), *rk1 = b.take<unsigned long long>(rows_cap);
    uint32_t *perm0 = b.take<uint32_t>(rows_cap), *perm1 = b.take<uint32_t>(rows_cap);
    char *cub_tmp2 = b.take<char>(cub_sort_rows);
    if (n_pairs + n_ring_rows) hipLaunchKernelGGL(k_expand_rows, grid(n_pairs + n_ring_rows, 256), dim3(256), 0, st, pairs_dev, (uint32_t)n_pairs, (const uint32_t *)first, rows,
                                                  (uint32_t)rows_cap, (const uint4 *)ring_rows, n_ring_rows, n_atom_rows);  // (the ring rows behind the atom rows)
    // The sort: stable radix passes over row indices, least significant key first.  The entity ranks, interaction, chains and model go
    // into ONE key when their actual widths fit 64 bits (else two or three passes); rows whose ten keys tie are put in order afterwards
    // (k_tie_fix).  `long_way`: the tie-breaking keys -- insertion codes (skipped when no atom carries one), distance -- as passes of their own.
    auto sort_and_finish = [&](bool long_way) -> arp_status {
        const uint32_t model_bits = width(std::max<uint64_t>(ds.n_models, 1)), top = model_bits + 2 * tb.chain_bits;
        int plan[5], end_bit[5], n_pass = 0;
        if (long_way) {
            plan[n_pass] = 0; end_bit[n_pass++] = 32;
            if (ds.any_icode) { plan[n_pass] = 1; end_bit[n_pass++] = 64; }
        }
        if (top + 2 * tb.rank_bits + 5 <= 64) { plan[n_pass] = 6; end_bit[n_pass++] = (int)(top + 2 * tb.rank_bits + 5); }
        else {
            plan[n_pass] = 2; end_bit[n_pass++] = (int)(tb.rank_bits + 5);
            if (top + tb.rank_bits <= 64) { plan[n_pass] = 5; end_bit[n_pass++] = (int)(top + tb.rank_bits); }
            else { plan[n_pass] = 3; end_bit[n_pass++] = (int)tb.rank_bits; plan[n_pass] = 4; end_bit[n_pass++] = (int)top; }
        }
        uint32_t *pin_ = perm0, *pout = perm1;
        for (int q = 0; q < n_pass; q++) {
            const int pass = plan[q];
            hipLaunchKernelGGL(k_row_key, grid(n_rows, 256), dim3(256), 0, st, n_rows, (const uint4 *)rows, q ? (const uint32_t *)pin_ : (const uint32_t *)nullptr, pass, (uint32_t)n,
                               (const EntKey *)ds.ent_key, (const EntKey *)d_ring_keys, (const uint32_t *)ent_rank, (const uint32_t *)ds.chain_rank, (const uint32_t *)ds.model,
                               (const uint32_t *)ds.model_rank, (const RingEnt *)d_rings, tb, rk0, q ? (uint32_t *)nullptr : pin_);
            size_t tmp = cub_sort_rows;
            TRY_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp2, tmp, (const unsigned long long *)rk0, rk1, (const uint32_t *)pin_, pout, (int)n_rows, 0, end_bit[q], st));
            std::swap(pin_, pout);
        }
        if (!long_way) {
            const bool one_key = plan[n_pass - 1] == 6 && n_pass == 1;  // rk1 then holds every row's ten keys, in sorted order
            hipLaunchKernelGGL(k_tie_fix, grid(n_rows, 256), dim3(256), 0, st, n_rows, (const uint4 *)rows, (const uint32_t *)pin_, one_key ? (const unsigned long long *)rk1 : nullptr,
                               (uint32_t)n, (const EntKey *)ds.ent_key, (const EntKey *)d_ring_keys, (const uint32_t *)ent_rank, (const uint32_t *)ds.chain_rank, (const uint32_t *)ds.model,
                               (const uint32_t *)ds.model_rank, (const RingEnt *)d_rings, pout, counters + 1);
            std::swap(pin_, pout);
        }
        hipLaunchKernelGGL(k_finish_rows, grid(n_rows, 256), dim3(256), 0, st, n_rows, (const uint4 *)rows, (const uint32_t *)pin_, (uint32_t)n, (const uint32_t *)ds.atom_sc_src,
                           (const RingEnt *)d_rings, (const PlaneD *)sc_pl, (const uint8_t *)valid, out_rows, out_sc, (const uint32_t *)(counters + 1), out_flag);
        TRY_HIP(hipGetLastError());
        return ARP_OK;
    };
    if (n_rows && (s = sort_and_finish(false)) != ARP_OK) return s;
    lap("sort+finish");
    // Rows, sc values and the tie flag come back in ONE copy.  A large table lands in a pooled pinned block that the table object then owns (the
    // Arrow export and the column accessors read it in place); a small one goes through the context's landing buffer into plain arrays.
    out->n = n_rows;
    const size_t row_bytes = (size_t)n_rows * 16, all_bytes = 2 * row_bytes + 16;
    char *heap = nullptr, *landing = nullptr;
    if (row_bytes >= (1u << 20)) {
        out->owner = pinned_block(all_bytes + 256);
        if (!out->owner) { set_error("out of pinned host memory for the table"); return ARP_ERR_OOM; }
        landing = out->owner.get();
    } else {
        heap = (char *)malloc(all_bytes + 64);
        if (!heap) { set_error("out of host memory"); return ARP_ERR_OOM; }
        out->owner = std::shared_ptr<char>(heap, [](char *q) { free(q); });
        landing = pin;
    }
    out->rows = reinterpret_cast<TableRow *>(heap ? heap : landing);
    out->sc = reinterpret_cast<TableSc *>((heap ? heap : landing) + row_bytes);
    uint32_t tie_overflow = 0u;
    auto fetch = [&]() -> arp_status {
        if (!n_rows) return ARP_OK;
        TRY_HIP(hipMemcpyAsync(landing, out_all, all_bytes, hipMemcpyDeviceToHost, st));
        TRY_HIP(hipStreamSynchronize(st));
        memcpy(&tie_overflow, landing + 2 * row_bytes, sizeof tie_overflow);
        if (heap) memcpy(heap, landing, 2 * row_bytes);
        return ARP_OK;
    };
    if ((s = fetch()) != ARP_OK) return s;
    if (tie_overflow) {  // a run of more than kTieRun rows with the same ten keys: sort again with the tie-breaking keys as passes of their own
        if ((s = sort_and_finish(true)) != ARP_OK || (s = fetch()) != ARP_OK) return s;
    }
    lap("unpack");
    return ARP_OK;
}

}  // namespace arp
