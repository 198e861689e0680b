// HIP kernels of the contact engine, written for gfx950 (MI355X, wave64) only.
//
// Pipeline (all on one stream, no host round trip):
//   k_init -> k_bounds -> k_setup -> k_zero_cells -> k_cellid -> scan(cell_count) -> k_scatter -> k_gather
//   -> k_pairs<COUNT> -> scan(task_count) -> k_pairs<FILL>
// It replaces the reference's R*-tree build + serial neighbour walk + rayon classification
// (src/contacts/complex.rs:189-299) with a uniform-grid cell list and a count/scan/fill pair emitter whose
// output order is deterministic.  Decisions are made in f64 with the reference's operation order and no FMA
// contraction (this file is compiled with -ffp-contract=off); an f32 test with a proven margin only prefilters.
#include "arp_internal.h"

namespace arp {

#define DEVFN __device__ __forceinline__

// ---------------------------------------------------------------------------------------------- helpers
DEVFN unsigned long long enc_f64(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
DEVFN double dec_f64(unsigned long long e) {
    unsigned long long u = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
    return __longlong_as_double((long long)u);
}
DEVFN uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane
DEVFN uint32_t mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
DEVFN double sq_dist(double ax, double ay, double az, double bx, double by, double bz) {
    // pdbtbx Atom::distance before the sqrt: (bx-ax)^2 + (by-ay)^2 + (bz-az)^2, left to right, no FMA
    double dx = __dsub_rn(bx, ax), dy = __dsub_rn(by, ay), dz = __dsub_rn(bz, az);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// ---------------------------------------------------------------------------------------------- grid build
__global__ void k_init(Bounds *b, unsigned long long *result) {
    if (threadIdx.x < 3) { b->mn[threadIdx.x] = ~0ull; b->mx[threadIdx.x] = 0ull; }
    if (threadIdx.x == 0) { b->n_models = 0; b->bad = 0; result[0] = 0; result[1] = 0; }
}

__global__ __launch_bounds__(256) void k_bounds(DevAtoms in, Bounds *b) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t models = 0, bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < in.n; i += gridDim.x * blockDim.x) {
        if (in.attr[i] & ARP_ATTR_H) continue;
        double p[3] = {in.x[i], in.y[i], in.z[i]};
        for (int k = 0; k < 3; k++) {
            if (!isfinite(p[k])) bad = 1;
            mn[k] = fmin(mn[k], p[k]);
            mx[k] = fmax(mx[k], p[k]);
        }
        models = max(models, (uint32_t)in.model[i] + 1u);
    }
    for (int off = 32; off; off >>= 1) {
        for (int k = 0; k < 3; k++) {
            mn[k] = fmin(mn[k], __shfl_xor(mn[k], off));
            mx[k] = fmax(mx[k], __shfl_xor(mx[k], off));
        }
        models = max(models, (uint32_t)__shfl_xor((int)models, off));
        bad |= (uint32_t)__shfl_xor((int)bad, off);
    }
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; k++) {
            if (mn[k] <= mx[k]) { atomicMin(&b->mn[k], enc_f64(mn[k])); atomicMax(&b->mx[k], enc_f64(mx[k])); }
        }
        atomicMax(&b->n_models, models);
        if (bad) atomicOr(&b->bad, 1u);
    }
}

__global__ void k_setup(const Bounds *b, GridParams *g, DevParams *prm, double cutoff, uint32_t ncells_cap) {
    if (threadIdx.x | blockIdx.x) return;
    double lo[3], ext[3];
    bool empty = b->mn[0] == ~0ull;
    for (int k = 0; k < 3; k++) {
        lo[k] = empty ? 0.0 : dec_f64(b->mn[k]);
        ext[k] = empty ? 0.0 : dec_f64(b->mx[k]) - lo[k];
        if (!(ext[k] >= 0.0) || !isfinite(ext[k])) ext[k] = 0.0;
        if (!isfinite(lo[k])) lo[k] = 0.0;
    }
    uint32_t nm = b->n_models ? b->n_models : 1u;
    // edge slightly above the cutoff so that |dx| <= cutoff can never straddle two cell boundaries after rounding
    double edge = cutoff * (1.0 + 1e-6);
    if (!(edge > 1e-3)) edge = 1e-3;
    double nx, ny, nz;
    for (;;) {
        nx = floor(ext[0] / edge) + 1.0; ny = floor(ext[1] / edge) + 1.0; nz = floor(ext[2] / edge) + 1.0;
        if (nx * ny * (nz + 1.0) * (double)nm <= (double)ncells_cap) break;
        edge *= 1.2599210498948732;  // sparse / huge extents: coarser cells stay correct (edge >= cutoff)
    }
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2];
    g->inv_edge = 1.0 / edge;
    g->nx = (uint32_t)nx; g->ny = (uint32_t)ny; g->nz = (uint32_t)nz;
    g->nzt = nm * (g->nz + 1u);
    g->ncells = g->nx * g->ny * g->nzt;
    g->bad = b->bad;
    // f32 prefilter: relative coordinates carry <= 2^-24 * extent of rounding each; a 10x-safe bound on the
    // induced error of dx^2+dy^2+dz^2 near the cutoff (derivation in DESIGN.md "Prefilter margin")
    double M = fmax(ext[0], fmax(ext[1], ext[2])) + edge;
    double margin = 4e-6 * (prm->r2 + fabs(cutoff) * M) + 1e-6;
    g->prefilter_margin = (float)margin;
    prm->r2f = __double2float_ru(prm->r2 + margin);
}

__global__ __launch_bounds__(256) void k_zero_cells(const GridParams *g, uint32_t *cell_count, uint32_t *cell_fill) {
    uint32_t n = g->ncells;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        cell_count[i] = 0;
        if (i < n) cell_fill[i] = 0;
    }
}

DEVFN uint32_t cell_index(const GridParams &g, double x, double y, double z, uint32_t model) {
    double fx = (x - g.ox) * g.inv_edge, fy = (y - g.oy) * g.inv_edge, fz = (z - g.oz) * g.inv_edge;
    uint32_t cx = (fx >= 0.0) ? (uint32_t)fmin(fx, 4.0e9) : 0u;  // NaN -> 0
    uint32_t cy = (fy >= 0.0) ? (uint32_t)fmin(fy, 4.0e9) : 0u;
    uint32_t cz = (fz >= 0.0) ? (uint32_t)fmin(fz, 4.0e9) : 0u;
    cx = min(cx, g.nx - 1u); cy = min(cy, g.ny - 1u); cz = min(cz, g.nz - 1u);
    uint32_t layer = model * (g.nz + 1u) + cz;  // every model owns a z slab followed by one empty layer
    layer = min(layer, g.nzt - 1u);
    return (layer * g.ny + cy) * g.nx + cx;
}

__global__ __launch_bounds__(256) void k_cellid(DevAtoms in, const GridParams *gp, uint32_t *cell_of_atom, uint32_t *cell_count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n) return;
    GridParams g = *gp;
    uint32_t c = ARP_NONE;
    if (!(in.attr[i] & ARP_ATTR_H)) {
        c = cell_index(g, in.x[i], in.y[i], in.z[i], in.model[i]);
        atomicAdd(&cell_count[c], 1u);
    }
    cell_of_atom[i] = c;
}

// ---------------------------------------------------------------------------------------------- scan
// Exclusive scan of in[0..n) (n read from device memory) into out[0..n], out[n] = total.  Three launches over a
// fixed 1024-block decomposition so that no host knowledge of n is needed.
constexpr uint32_t kScanBlocks = 1024, kScanThreads = 256;

template <typename TOut>
DEVFN TOut block_exclusive_scan(TOut v, TOut *total, TOut *lds /* [kScanThreads/64 + 1] */) {
    // wave inclusive scan
    TOut inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        TOut t = __shfl_up(inc, off);
        if ((threadIdx.x & 63) >= (uint32_t)off) inc += t;
    }
    uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) lds[w] = inc;
    __syncthreads();
    TOut wave_off = 0, tot = 0;
    for (uint32_t k = 0; k < kScanThreads / 64; k++) { TOut s = lds[k]; if (k < w) wave_off += s; tot += s; }
    *total = tot;
    return wave_off + inc - v;
}

template <typename TOut>
__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const uint32_t *in, const uint32_t *n_ptr, TOut *tmp) {
    __shared__ TOut lds[kScanThreads / 64 + 1];
    uint32_t n = *n_ptr;
    uint32_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
    uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
    TOut s = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kScanThreads) s += in[i];
    TOut tot;
    block_exclusive_scan<TOut>(s, &tot, lds);
    if (threadIdx.x == 0) tmp[blockIdx.x] = tot;
}
template <typename TOut>
__global__ __launch_bounds__(kScanThreads) void k_scan_tmp(TOut *tmp) {  // tmp[kScanBlocks] receives the grand total
    __shared__ TOut lds[kScanThreads / 64 + 1];
    TOut carry = 0;
    for (uint32_t base = 0; base < kScanBlocks; base += kScanThreads) {
        TOut v = tmp[base + threadIdx.x], tot;
        TOut ex = block_exclusive_scan<TOut>(v, &tot, lds);
        tmp[base + threadIdx.x] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) tmp[kScanBlocks] = carry;
}
template <typename TOut>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const uint32_t *in, const uint32_t *n_ptr, const TOut *tmp, TOut *out) {
    __shared__ TOut lds[kScanThreads / 64 + 1];
    uint32_t n = *n_ptr;
    uint32_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
    uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
    TOut carry = tmp[blockIdx.x];
    for (uint32_t base = lo; base < hi; base += kScanThreads) {
        uint32_t i = base + threadIdx.x;
        TOut v = (i < hi) ? (TOut)in[i] : (TOut)0, tot;
        TOut ex = block_exclusive_scan<TOut>(v, &tot, lds);
        if (i < hi) out[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tmp[kScanBlocks];
}

// ---------------------------------------------------------------------------------------------- sort into cells
__global__ __launch_bounds__(256) void k_scatter(uint32_t n, const uint32_t *cell_of_atom, const uint32_t *cell_start,
                                                 uint32_t *cell_fill, uint32_t *perm, uint32_t *slot_cell) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = cell_of_atom[i];
    if (c == ARP_NONE) return;
    uint32_t p = cell_start[c] + atomicAdd(&cell_fill[c], 1u);
    perm[p] = i;
    slot_cell[p] = c;
}

// Final slot = cell_start + rank of the atom index inside its cell: the sorted order (and therefore the order of
// the emitted pairs) does not depend on the arrival order of the atomics above.
__global__ __launch_bounds__(256) void k_gather(DevAtoms in, const GridParams *gp, const uint32_t *cell_start,
                                                const uint32_t *perm, const uint32_t *slot_cell, Sorted so) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    GridParams g = *gp;
    uint32_t n_heavy = cell_start[g.ncells];
    if (p >= n_heavy) return;
    uint32_t c = slot_cell[p], i = perm[p];
    uint32_t s = cell_start[c], e = cell_start[c + 1], rank = 0;
    for (uint32_t q = s; q < e; q++) rank += (perm[q] < i) ? 1u : 0u;
    uint32_t d = s + rank;
    double x = in.x[i], y = in.y[i], z = in.z[i];
    so.x[d] = x; so.y[d] = y; so.z[d] = z;
    so.rec[d] = make_float4((float)(x - g.ox), (float)(y - g.oy), (float)(z - g.oz), __uint_as_float(i));
    so.meta[d] = make_uint4(in.attr[i], in.res_ord[i], (uint32_t)in.chain_rank[i] | ((uint32_t)in.model[i] << 16), i);
}

// ---------------------------------------------------------------------------------------------- per-pair rules
// should_compare_entities(x, y, symmetric = true) for x in L, y in R (complex.rs:76-131, 200-206); hydrogens never
// reach here (they are not in the grid).
DEVFN bool candidate(uint4 mx, uint4 my) {
    if (!(mx.x & ARP_ATTR_LIGAND) || !(my.x & ARP_ATTR_RECEPTOR)) return false;
    if ((mx.z >> 16) != (my.z >> 16)) return false;                      // :96-98 same model
    if ((mx.z & 0xFFFFu) == (my.z & 0xFFFFu))                            // :108 same chain
        return (my.y > 1u) && (mx.y < my.y - 1u);                        // :113
    bool both = (mx.x & my.x & ARP_ATTR_LIGAND) && (mx.x & my.x & ARP_ATTR_RECEPTOR);
    return !(both && ((mx.z & 0xFFFFu) > (my.z & 0xFFFFu)));             // :124-129
}

DEVFN double angle_deg(const double a[3], const double b[3], const double c[3]) {
    // pdbtbx Atom::angle: angle at b between b->a and b->c, degrees
    double ba[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, bc[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    double nba = sqrt(0.0 + ba[0] * ba[0] + ba[1] * ba[1] + ba[2] * ba[2]);
    double nbc = sqrt(0.0 + bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
    double dot = 0.0 + ba[0] * bc[0] + ba[1] * bc[1] + ba[2] * bc[2];
    return acos(dot / (nba * nbc)) * (180.0 / 3.14159265358979323846264338327950288);
}
DEVFN double dihedral_deg(const double a[3], const double b[3], const double c[3], const double d[3]) {
    double ba[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, bc[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    double cb[3] = {b[0] - c[0], b[1] - c[1], b[2] - c[2]}, cd[3] = {d[0] - c[0], d[1] - c[1], d[2] - c[2]};
    double n1[3] = {ba[1] * bc[2] - ba[2] * bc[1], ba[2] * bc[0] - ba[0] * bc[2], ba[0] * bc[1] - ba[1] * bc[0]};
    double n2[3] = {cb[1] * cd[2] - cb[2] * cd[1], cb[2] * cd[0] - cb[0] * cd[2], cb[0] * cd[1] - cb[1] * cd[0]};
    double a1 = sqrt(0.0 + n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    double a2 = sqrt(0.0 + n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
    double dot = 0.0 + n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
    return acos(dot / (a1 * a2)) * (180.0 / 3.14159265358979323846264338327950288);
}
DEVFN void load_pos(const DevAtoms &in, uint32_t i, double p[3]) { p[0] = in.x[i]; p[1] = in.y[i]; p[2] = in.z[i]; }

// hbond.rs:36-63 / 80-107 for a fixed (donor, acceptor) assignment.  Returns 2 = (weak) hydrogen bond,
// 1 = (weak) polar contact, 0 = nothing.
DEVFN int hbond_like(const DevAtoms &in, const DevParams *prm, double s, uint32_t donor_idx, const double pd[3],
                     const double pa[3], uint32_t acc_attr, double min_angle) {
    if (s < prm->s_ion && in.n_res) {  // da_dist <= 4.0: probe every hydrogen of the donor's residue (hbond.rs:38-42)
        uint32_t r = in.res_id[donor_idx];
        uint32_t p0 = in.res_h_ptr[r], p1 = in.res_h_ptr[r + 1];
        double lim = prm->s_hacc[acc_attr & ARP_ATTR_ELEM_MASK];
        for (uint32_t p = p0; p < p1; p++) {
            double ph[3];
            load_pos(in, in.res_h_idx[p], ph);
            if (sq_dist(ph[0], ph[1], ph[2], pa[0], pa[1], pa[2]) < lim && angle_deg(pd, ph, pa) >= min_angle) return 2;
        }
    }
    return (s < prm->s_polar) ? 1 : 0;
}

// All rows of one candidate pair as a bit set (complex.rs:217-296).  x = ligand, y = receptor.
DEVFN uint32_t classify(const DevAtoms &in, const DevParams *prm, double s, uint4 mx, uint4 my, const double px[3],
                        const double py[3], unsigned long long *result) {
    const uint32_t ax = mx.x, ay = my.x;
    const uint32_t e = ((ax & ARP_ATTR_ELEM_MASK) << 4) | (ay & ARP_ATTR_ELEM_MASK);
    uint32_t kind = 0;
    // vdw.rs:32-43
    if (s < prm->s_clash[e]) return 1u << ARP_StericClash;  // complex.rs:233-235: nothing else is looked at
    if (s < prm->s_cov[e]) {
        bool ss = false;
        if ((ax & ay & ARP_ATTR_CYS_SG) && in.n_res) {  // vdw.rs:46-80
            uint32_t r1 = in.res_id[mx.w], r2 = in.res_id[my.w];
            uint32_t cb1 = in.res_cb[r1], s1 = in.res_sg[r1], s2 = in.res_sg[r2], cb2 = in.res_cb[r2];
            if (cb1 == ARP_NONE || cb2 == ARP_NONE || s1 == ARP_NONE || s2 == ARP_NONE) {
                atomicOr(&result[1], 2ull);  // the reference unwrap()s and panics here
            } else {
                double a[3], b[3], c[3], d[3];
                load_pos(in, cb1, a); load_pos(in, s1, b); load_pos(in, s2, c); load_pos(in, cb2, d);
                double dih = fabs(dihedral_deg(a, b, c, d));
                ss = (dih >= 60.0) && (dih <= 120.0);
            }
        }
        kind |= 1u << (ss ? ARP_Disulfide : ARP_CovalentBond);
    } else if (s < prm->s_vdw[e]) {
        kind |= 1u << ARP_VanDerWaalsContact;
    }
    const bool near4 = s < prm->s_ion;  // d <= 4.0
    // ionic.rs:11-22,37-57
    const bool ionic = near4 && (((ax & ARP_ATTR_POS) && (ay & ARP_ATTR_NEG)) || ((ay & ARP_ATTR_POS) && (ax & ARP_ATTR_NEG)));
    // hbond.rs:30-66,113-134: (e1 donor, e2 acceptor) is tried first
    int hb = 0;
    if ((ax & ARP_ATTR_DONOR) && (ay & ARP_ATTR_ACCEPTOR)) hb = hbond_like(in, prm, s, mx.w, px, py, ay, 90.0);
    else if ((ay & ARP_ATTR_DONOR) && (ax & ARP_ATTR_ACCEPTOR)) hb = hbond_like(in, prm, s, my.w, py, px, ax, 90.0);
    // complex.rs:240-251
    if (ionic) kind |= 1u << (hb == 2 ? ARP_SaltBridge : ARP_IonicBond);
    else if (hb) kind |= 1u << (hb == 2 ? ARP_HydrogenBond : ARP_PolarContact);
    // hbond.rs:74-110,181-201
    int wk = 0;
    if ((ax & ARP_ATTR_WEAK_DONOR) && (ay & ARP_ATTR_ACCEPTOR)) wk = hbond_like(in, prm, s, mx.w, px, py, ay, 130.0);
    else if ((ay & ARP_ATTR_WEAK_DONOR) && (ax & ARP_ATTR_ACCEPTOR)) wk = hbond_like(in, prm, s, my.w, py, px, ax, 130.0);
    if (wk) kind |= 1u << (wk == 2 ? ARP_WeakHydrogenBond : ARP_WeakPolarContact);
    // ionic.rs:25-35,59-81
    if (near4 && ((ax & ay & ARP_ATTR_POS) || (ax & ay & ARP_ATTR_NEG))) kind |= 1u << ARP_IonicRepulsion;
    // hydrophobic.rs:10-24
    if ((ax & ay & ARP_ATTR_HYDROPHOBIC) && s < prm->s_hphob) kind |= 1u << ARP_HydrophobicContact;
    return kind;
}

// ---------------------------------------------------------------------------------------------- pair search
// One wave per home cell.  Half shell: the home cell against itself (slot order breaks the tie), its +x neighbour,
// the three cells of row (y+1, z) and the nine cells of layer z+1 -- five contiguous slot ranges because cells are
// x-major.  Every unordered pair is therefore tested exactly once; the reference's ordered pair (x in L, y in R)
// is recovered by candidate(), of which at most one orientation can hold (complex.rs:108-130).
//
// Phase 1 (all lanes): flat (home, neighbour) enumeration, f32 distance prefilter, survivors are compacted into an
// LDS queue with a wavefront ballot + prefix count.  Phase 2 (full waves of 64 survivors): exact f64 decision,
// classification, and either a count (COUNT pass) or a coalesced 16-byte-per-lane store (FILL pass).
constexpr int kWavesPerBlock = 4;
constexpr int kQueue = 128;
constexpr uint32_t kPairBlocks = 256 * 8;  // persistent: 8 blocks of 4 waves per CU, cells dealt round-robin

template <bool FILL>
DEVFN uint32_t process_batch(const DevAtoms &in, const DevParams *prm, const Sorted &so, uint2 ent, bool active,
                             unsigned long long base, uint32_t emitted, arp_pair *out, unsigned long long capacity,
                             unsigned long long *result) {
    bool valid = false, swap = false;
    double s = 0.0;
    uint4 ma = make_uint4(0, 0, 0, 0), mb = ma;
    double pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
    if (active) {
        pa[0] = so.x[ent.x]; pa[1] = so.y[ent.x]; pa[2] = so.z[ent.x];
        pb[0] = so.x[ent.y]; pb[1] = so.y[ent.y]; pb[2] = so.z[ent.y];
        s = sq_dist(pa[0], pa[1], pa[2], pb[0], pb[1], pb[2]);
        if (s <= prm->r2) {  // rstar: inclusive
            ma = so.meta[ent.x]; mb = so.meta[ent.y];
            if (candidate(ma, mb)) valid = true;
            else if (candidate(mb, ma)) { valid = true; swap = true; }
        }
    }
    unsigned long long vm = __ballot(valid);
    if (FILL) {
        if (valid) {
            uint32_t kind = swap ? classify(in, prm, s, mb, ma, pb, pa, result) : classify(in, prm, s, ma, mb, pa, pb, result);
            unsigned long long pos = base + emitted + mbcnt(vm);
            if (pos < capacity) {
                arp_pair r;
                r.i = swap ? mb.w : ma.w; r.j = swap ? ma.w : mb.w;
                r.dist = (float)sqrt(s);
                r.kind = kind;
                out[pos] = r;
            }
        }
    }
    return (uint32_t)__popcll(vm);
}

template <bool FILL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_pairs(DevAtoms in, const GridParams *gp, const DevParams *prm,
                                                               const uint32_t *cell_start, Sorted so, uint32_t *task_count,
                                                               const unsigned long long *task_base, arp_pair *out,
                                                               unsigned long long capacity, unsigned long long *result) {
    __shared__ uint2 queue[kWavesPerBlock][kQueue];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, ncells = gp->ncells;
    const float r2f = prm->r2f;
    uint2 *q = queue[wave];
#pragma unroll 1
    for (uint32_t c = blockIdx.x * kWavesPerBlock + wave; c < ncells; c += gridDim.x * kWavesPerBlock) {
    const uint32_t hs = cell_start[c], he = cell_start[c + 1], nh = he - hs;
    if (nh == 0) { if (!FILL && lane == 0) task_count[c] = 0; continue; }
    const uint32_t cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
    const uint32_t xlo = cx ? cx - 1 : 0, xhi = min(cx + 1, nx - 1);
    // five slot ranges
    uint32_t rs[5], re[5];
    {
        uint32_t row = (cz * ny + cy) * nx;
        rs[0] = hs; re[0] = cell_start[row + xhi + 1];
        int k = 1;
        for (int dz = 0; dz <= 1; dz++)
            for (int dy = (dz ? -1 : 1); dy <= 1; dy++, k++) {
                int yy = (int)cy + dy; uint32_t zz = cz + dz;
                if (yy < 0 || yy >= (int)ny || zz >= nzt) { rs[k] = re[k] = 0; continue; }
                uint32_t r = (zz * ny + (uint32_t)yy) * nx;
                rs[k] = cell_start[r + xlo]; re[k] = cell_start[r + xhi + 1];
            }
    }
    const unsigned long long base = FILL ? task_base[c] : 0ull;
    uint32_t qlen = 0, emitted = 0;
    const uint32_t dh = 64u % nh, dn = 64u / nh;
#pragma unroll 1
    for (int k = 0; k < 5; k++) {
        const uint32_t ns = rs[k], nn = re[k] - rs[k];
        if (nn == 0) continue;
        uint32_t hoff = lane % nh, noff = lane / nh;
        const unsigned long long total = (unsigned long long)nh * nn;
#pragma unroll 1
        for (unsigned long long t0 = 0; t0 < total; t0 += 64) {
            bool pass = false;
            uint32_t h = hs + hoff, n = ns + noff;
            if (noff < nn) {
                float4 a = so.rec[h], b = so.rec[n];
                float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
                float d2 = dx * dx + dy * dy + dz * dz;
                pass = (d2 <= r2f) && (k != 0 || n > h);
            }
            unsigned long long m = __ballot(pass);
            if (pass) q[qlen + mbcnt(m)] = make_uint2(h, n);
            qlen += (uint32_t)__popcll(m);
            if (qlen >= 64) {
                qlen -= 64;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // lanes read entries other lanes wrote
                uint2 ent = q[qlen + lane];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                emitted += process_batch<FILL>(in, prm, so, ent, true, base, emitted, out, capacity, result);
            }
            hoff += dh; noff += dn;
            if (hoff >= nh) { hoff -= nh; noff++; }
        }
    }
    if (qlen) {
        bool act = lane < qlen;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint2 ent = act ? q[lane] : make_uint2(0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        emitted += process_batch<FILL>(in, prm, so, ent, act, base, emitted, out, capacity, result);
    }
    if (!FILL && lane == 0) task_count[c] = emitted;
    }  // cell loop
}

__global__ void k_finish(const GridParams *g, const unsigned long long *task_base, unsigned long long *result, unsigned long long capacity,
                         int have_out) {
    if (threadIdx.x | blockIdx.x) return;
    unsigned long long total = task_base[g->ncells];
    result[0] = total;
    if (have_out && total > capacity) result[1] |= 1ull;
    if (g->bad) result[1] |= 4ull;
}

// ---------------------------------------------------------------------------------------------- profiler + launch
void Profiler::begin(const char *name, hipStream_t st) {
    if (!enabled) return;
    if (!created) { for (int k = 0; k < kMax; k++) { (void)hipEventCreate(&ev0[k]); (void)hipEventCreate(&ev1[k]); } created = true; }
    if (n >= kMax) return;
    names[n] = name;
    (void)hipEventRecord(ev0[n], st);
}
void Profiler::end(hipStream_t st) {
    if (!enabled || n >= kMax) return;
    (void)hipEventRecord(ev1[n], st);
    n++;
}

template <typename TOut>
static void launch_scan(const uint32_t *in, const uint32_t *n_ptr, TOut *tmp, TOut *out, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_reduce<TOut>, dim3(kScanBlocks), dim3(kScanThreads), 0, st, in, n_ptr, tmp);
    hipLaunchKernelGGL(k_scan_tmp<TOut>, dim3(1), dim3(kScanThreads), 0, st, tmp);
    hipLaunchKernelGGL(k_scan_apply<TOut>, dim3(kScanBlocks), dim3(kScanThreads), 0, st, in, n_ptr, (const TOut *)tmp, out);
}

void launch_pipeline(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, bool fill,
                     Profiler *prof, double cutoff) {
    const uint32_t n = in.n;
    const uint32_t nb = (n + 255) / 256;
    auto P0 = [&](const char *nm) { if (prof) prof->begin(nm, st); };
    auto P1 = [&]() { if (prof) prof->end(st); };
    if (prof) prof->n = 0;
    P0("grid_bounds");
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, st, ws.bounds, ws.result);
    if (n) hipLaunchKernelGGL(k_bounds, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, in, ws.bounds);
    hipLaunchKernelGGL(k_setup, dim3(1), dim3(1), 0, st, (const Bounds *)ws.bounds, ws.grid, ws.params, cutoff, ws.ncells_cap);
    P1();
    P0("grid_count");
    hipLaunchKernelGGL(k_zero_cells, dim3(1024), dim3(256), 0, st, (const GridParams *)ws.grid, ws.cell_count, ws.cell_fill);
    if (n) hipLaunchKernelGGL(k_cellid, dim3(nb), dim3(256), 0, st, in, (const GridParams *)ws.grid, ws.cell_of_atom, ws.cell_count);
    P1();
    P0("grid_scan");
    launch_scan<uint32_t>(ws.cell_count, &ws.grid->ncells, ws.scan_tmp, ws.cell_start, st);
    P1();
    P0("grid_sort");
    if (n) {
        hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, st, n, (const uint32_t *)ws.cell_of_atom, (const uint32_t *)ws.cell_start,
                           ws.cell_fill, ws.perm, ws.slot_cell);
        hipLaunchKernelGGL(k_gather, dim3(nb), dim3(256), 0, st, in, (const GridParams *)ws.grid, (const uint32_t *)ws.cell_start,
                           (const uint32_t *)ws.perm, (const uint32_t *)ws.slot_cell, ws.sorted);
    }
    P1();
    const uint32_t pair_blocks = kPairBlocks;
    P0("pairs_count");
    hipLaunchKernelGGL(k_pairs<false>, dim3(pair_blocks), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid,
                       (const DevParams *)ws.params, (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count,
                       (const unsigned long long *)ws.task_base, (arp_pair *)nullptr, 0ull, ws.result);
    P1();
    P0("pairs_scan");
    launch_scan<unsigned long long>(ws.task_count, &ws.grid->ncells, ws.scan_tmp64, ws.task_base, st);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(1), 0, st, (const GridParams *)ws.grid, (const unsigned long long *)ws.task_base, ws.result,
                       capacity, fill ? 1 : 0);
    P1();
    if (fill) {
        P0("pairs_fill");
        hipLaunchKernelGGL(k_pairs<true>, dim3(pair_blocks), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid,
                           (const DevParams *)ws.params, (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count,
                           (const unsigned long long *)ws.task_base, out, capacity, ws.result);
        P1();
    }
}

void launch_fill_only(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof) {
    const uint32_t pair_blocks = kPairBlocks;
    if (prof) prof->begin("pairs_fill", st);
    hipLaunchKernelGGL(k_pairs<true>, dim3(pair_blocks), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid,
                       (const DevParams *)ws.params, (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count,
                       (const unsigned long long *)ws.task_base, out, capacity, ws.result);
    if (prof) prof->end(st);
}

}  // namespace arp
