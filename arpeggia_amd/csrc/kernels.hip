// HIP kernels of the contact engine, written for gfx950 (MI355X, wave64) only.
//
// Pipeline (all on one stream, no host round trip):
//   k_bounds -> k_cellid (sizes the grid first) -> scan(cell_count) -> k_place | k_scatter + k_gather  (grid.inl)
//   -> k_emit -> probe pass -> k_fixup   |   k_pairs<count> -> scan -> k_pairs<fill>           (pairs_emit.inl, pairs.inl)
//   below 20 480 atoms: k_cellid (finds the box as well) -> k_scan_one -> k_place -> k_emit (hole-free list, probes inline)
// It replaces the reference's R*-tree build + serial neighbour walk + rayon classification
// (src/contacts/complex.rs:189-299) with a uniform-grid cell list and a count/scan/fill pair emitter whose
// output order is deterministic.  Decisions are made in f64 with the reference's operation order and no FMA
// contraction (this file is compiled with -ffp-contract=off); an f32 test with a proven margin only prefilters.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

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
// number of set bits of `mask` below this lane
DEVFN uint32_t mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
DEVFN double sq_dist(double ax, double ay, double az, double bx, double by, double bz) {
    // pdbtbx Atom::distance before the sqrt: (bx-ax)^2 + (by-ay)^2 + (bz-az)^2, left to right, no FMA
    double dx = __dsub_rn(bx, ax), dy = __dsub_rn(by, ay), dz = __dsub_rn(bz, az);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}
// Wave-wide min / max on the vector ALU's data-parallel primitives (DPP): a 16-lane inclusive scan by row shifts, then the
// row results hop to the later rows (row_bcast 15 / 31) and lane 63 holds the answer.  __shfl_xor compiles to six ds_bpermute
// round trips through the LDS crossbar, each waited for: ~1 us per reduction in a wave that has little else to overlap.
template <bool MAX>
DEVFN uint32_t wave_reduce_u32(uint32_t v) {
    const uint32_t id = MAX ? 0u : 0xFFFFFFFFu;
    auto op = [](uint32_t a, uint32_t b) { return MAX ? max(a, b) : min(a, b); };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
DEVFN uint32_t wave_min_u32(uint32_t v) { return wave_reduce_u32<false>(v); }
DEVFN uint32_t wave_max_u32(uint32_t v) { return wave_reduce_u32<true>(v); }
DEVFN void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// ---------------------------------------------------------------------------------------------- grid build, scan, sort
#include "grid.inl"

// ---------------------------------------------------------------------------------------------- per-pair rules
struct LdsParams {        // block-shared copy of the decision bounds (6.3 KB) + the pair-rule table (2 KB)
    double s_clash[256], s_cov[256], s_vdw[256], s_hacc[16];
    double r2, s_ion, s_polar, s_hphob;
    uint32_t contacts_only, all_both;
    uint32_t lut[512];    // pair_lut_entry(): rows decided by class bits and distance level alone
};

// should_compare_entities(x, y, symmetric = true) for x in L, y in R (complex.rs:76-131, 200-206), evaluated for both
// orientations of an unordered pair at once: 1 = (a ligand, b receptor), 2 = (b ligand, a receptor), 0 = not a candidate.
// At most one orientation can hold.  Hydrogens never reach here (they are not in the grid).  Written without branches:
// the exact phase is instruction-issue bound and every divergent branch costs a handful of scalar exec-mask instructions.
DEVFN int orient(const Fat &a, const Fat &b) {
    // (:96-98, the same-model test, is the grid's: every model owns its own slab of cells + an empty separator layer, so a slot window never
    // holds an atom of another model -- the chain rank has all 32 bits of crm)
    const bool same_model = true;
    const uint32_t ca = a.crm, cb = b.crm;
    const bool aL = a.pw & kPwLigand, aR = a.pw & kPwReceptor, bL = b.pw & kPwLigand, bR = b.pw & kPwReceptor;
    // same chain (:108,:113): (e2 > 1) && (e1 < e2 - 1)  <=>  e1 + 1 < e2   (ordinals are far below 2^32 - 1)
    const bool ab_chain = a.res_ord + 1u < b.res_ord, ba_chain = b.res_ord + 1u < a.res_ord;
    const bool both = aL & aR & bL & bR;                                 // :124-129
    const bool ab_cross = !(both & (ca > cb)), ba_cross = !(both & (cb > ca));
    const bool same_chain = ca == cb;
    const bool ab = same_chain ? ab_chain : ab_cross, ba = same_chain ? ba_chain : ba_cross;
    const bool o1 = same_model & aL & bR & ab, o2 = same_model & bL & aR & ba;
    return o1 ? 1 : (o2 ? 2 : 0);
}
// The same when every atom is in both sets (groups "/", GridParams::all_both): the L/R tests drop out, a same-chain pair is a
// candidate iff the ordinals differ by >= 2 (:113), a cross-chain pair of one model always (:124-129), and in both cases the
// ligand is the atom with the smaller (chain rank, ordinal) key.
DEVFN int orient_all_both(const Fat &a, const Fat &b) {
    const bool same_model = true, same_mc = a.crm == b.crm;  // (the model: see orient)
    const bool gap = (b.res_ord - a.res_ord + 1u) > 2u;                  // |ordinal difference| >= 2
    const bool valid = same_model & (!same_mc | gap);
    const unsigned long long ka = ((unsigned long long)a.crm << 32) | a.res_ord, kb = ((unsigned long long)b.crm << 32) | b.res_ord;
    return valid ? (kb < ka ? 2 : 1) : 0;
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

// hbond.rs:36-63 / 80-107: the hydrogen probe, out of line -- it runs only for donor residues that really carry hydrogens.
__device__ __noinline__ int hydrogen_probe(const double *X, const double *Y, const double *Z, const uint32_t *res_h_idx, double lim,
                                           uint32_t p0, uint32_t p1, double dx, double dy, double dz, double ax, double ay, double az,
                                           double min_angle) {
    const double pd[3] = {dx, dy, dz}, pa[3] = {ax, ay, az};
    // four hydrogens per trip: index loads, then twelve coordinate loads in flight together -- one by one the loop is a chain
    // of dependent round trips (index -> coordinates) per hydrogen.  "Some hydrogen qualifies" does not depend on the order.
    for (uint32_t p = p0; p < p1; p += 4u) {
        uint32_t h[4];
        double hx[4], hy[4], hz[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) h[u] = res_h_idx[min(p + u, p1 - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) { hx[u] = X[h[u]]; hy[u] = Y[h[u]]; hz[u] = Z[h[u]]; }
        bool hit = false;
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const double ph[3] = {hx[u], hy[u], hz[u]};
            if (p + u < p1 && sq_dist(ph[0], ph[1], ph[2], pa[0], pa[1], pa[2]) < lim) hit = hit || (angle_deg(pd, ph, pa) >= min_angle);
        }
        if (hit) return 1;
    }
    return 0;
}
// (weak) hydrogen bond / polar contact for the donor chosen by `donor_is_a`.  Returns 2 = hydrogen bond, 1 = polar contact, 0 = nothing.
// vdw.rs:46-80, out of line (rare): 1 = disulfide, 0 = plain covalent, sets the error flag where the reference panics
__device__ __noinline__ int disulfide_probe(const double *X, const double *Y, const double *Z, const uint32_t *res_id, const uint32_t *res_cb,
                                            const uint32_t *res_sg, uint32_t ix, uint32_t iy, unsigned long long *result) {
    uint32_t r1 = res_id[ix], r2 = res_id[iy];
    uint32_t cb1 = res_cb[r1], s1 = res_sg[r1], s2 = res_sg[r2], cb2 = res_cb[r2];
    if (cb1 == ARP_NONE || cb2 == ARP_NONE || s1 == ARP_NONE || s2 == ARP_NONE) {
        atomicOr(&result[1], 2ull);  // the reference unwrap()s and panics here
        return 0;
    }
    const double a[3] = {X[cb1], Y[cb1], Z[cb1]}, b[3] = {X[s1], Y[s1], Z[s1]}, c[3] = {X[s2], Y[s2], Z[s2]}, d[3] = {X[cb2], Y[cb2], Z[cb2]};
    double dih = fabs(dihedral_deg(a, b, c, d));
    return (dih >= 60.0) && (dih <= 120.0);
}

constexpr uint32_t kDeferKind = 0xFFFFFFFFu;  // classify<false>: the pair needs a hydrogen / disulfide probe, decide it in the deferred pass

// (weak) hydrogen bond test of hbond.rs:36-58 / 80-102 for the donor chosen by `donor_is_a`: true iff some hydrogen of the donor's
// residue satisfies the distance and angle conditions.  Only called for donors whose residue carries hydrogens.
// (P: LdsParams, or ProbeParamsE of the small-input emit kernels -- anything with the bounds as members)
template <typename P>
DEVFN bool hbond_probe(const DevAtoms &in, const P &prm, bool donor_is_a, const Fat &a, const Fat &b, double min_angle) {
    const uint32_t res = in.res_id[donor_is_a ? a.orig : b.orig];
    const uint2 hi = make_uint2(in.res_h_ptr[res], in.res_h_ptr[res + 1]);
    const uint32_t acc_attr = donor_is_a ? b.attr : a.attr;
    const double dx = donor_is_a ? a.x : b.x, dy = donor_is_a ? a.y : b.y, dz = donor_is_a ? a.z : b.z;
    const double ax = donor_is_a ? b.x : a.x, ay = donor_is_a ? b.y : a.y, az = donor_is_a ? b.z : a.z;
    return hydrogen_probe(in.x, in.y, in.z, in.res_h_idx, prm.s_hacc[acc_attr & ARP_ATTR_ELEM_MASK], hi.x, hi.y, dx, dy, dz, ax, ay, az, min_angle) != 0;
}

// All rows of one candidate pair as a bit set (complex.rs:217-296).  The rules are symmetric in the two atoms except for
// the order in which the two donor/acceptor assignments are tried (hbond.rs:125-133: the ligand as donor first) and the
// argument order of the disulfide dihedral; `swap` says that b is the ligand.  Everything on the common path is
// straight-line predicate arithmetic; only the rare probes branch (PROBES) or defer the pair (!PROBES).
template <bool PROBES, typename P = LdsParams>
DEVFN uint32_t classify(const DevAtoms &in, const P &prm, double s, const Fat &a, const Fat &b, bool swap, unsigned long long *result) {
    const uint32_t aa = a.attr, ab = b.attr, both = aa & ab;
    const uint32_t e = ((aa & ARP_ATTR_ELEM_MASK) << 4) | (ab & ARP_ATTR_ELEM_MASK);  // the radius tables are symmetric
    const double t_clash = prm.s_clash[e], t_cov = prm.s_cov[e], t_vdw = prm.s_vdw[e];
    const bool clash = s < t_clash, cov = s < t_cov, vdw = s < t_vdw;                 // vdw.rs:32-43 (strict <)
    const bool near4 = s < prm.s_ion, near35 = s < prm.s_polar, near45 = s < prm.s_hphob;  // d <= 4.0 / 3.5 / 4.5
    // attribute bit positions: DONOR 4, ACCEPTOR 5, WEAK_DONOR 6, POS 7, NEG 8, HYDROPHOBIC 9, CYS_SG 10, has-H 31
    const bool d_ab = (aa >> 4) & (ab >> 5) & 1u, d_ba = (ab >> 4) & (aa >> 5) & 1u;  // hbond.rs:113-134
    const bool w_ab = (aa >> 6) & (ab >> 5) & 1u, w_ba = (ab >> 6) & (aa >> 5) & 1u;  // hbond.rs:181-201
    const bool strong = d_ab | d_ba, weak = w_ab | w_ba;
    const bool sd_a = swap ? !d_ba : d_ab, wd_a = swap ? !w_ba : w_ab;                // the ligand is tried as donor first
    const bool s_hasH = (sd_a ? aa : ab) >> 31, w_hasH = (wd_a ? aa : ab) >> 31;
    const bool need_hs = strong & near4 & s_hasH, need_hw = weak & near4 & w_hasH;    // hbond.rs:37,81: da_dist <= 4.0
    const bool need_ss = cov & !clash & ((both >> 10) & 1u) & (in.n_res != 0u);       // vdw.rs:46-53
    bool hb2 = false, wk2 = false, ss = false;
    if (!clash && (need_hs | need_hw | need_ss)) {                                     // rare
        if (!PROBES) return kDeferKind;
        if (need_ss) ss = disulfide_probe(in.x, in.y, in.z, in.res_id, in.res_cb, in.res_sg, swap ? b.orig : a.orig, swap ? a.orig : b.orig, result);
        if (need_hs) hb2 = hbond_probe(in, prm, sd_a, a, b, 90.0);
        if (need_hw) wk2 = hbond_probe(in, prm, wd_a, a, b, 130.0);
    }
    const bool ionic = near4 & ((((aa >> 7) & (ab >> 8)) | ((ab >> 7) & (aa >> 8))) & 1u);     // ionic.rs:11-22,37-57
    const bool repel = near4 & (((both >> 7) | (both >> 8)) & 1u);                              // ionic.rs:25-35,59-81
    const bool hphob = near45 & ((both >> 9) & 1u);                                            // hydrophobic.rs:10-24
    const bool hb1 = strong & near35, wk1 = weak & near35;                                      // polar contacts (hbond.rs:60-63)
    uint32_t kind = cov ? (1u << (ss ? ARP_Disulfide : ARP_CovalentBond)) : (vdw ? (1u << ARP_VanDerWaalsContact) : 0u);
    // complex.rs:240-251
    const uint32_t electro = ionic ? (hb2 ? ARP_SaltBridge : ARP_IonicBond) : (hb2 ? ARP_HydrogenBond : ARP_PolarContact);
    kind |= (ionic | hb2 | hb1) ? (1u << electro) : 0u;
    kind |= wk2 ? (1u << ARP_WeakHydrogenBond) : (wk1 ? (1u << ARP_WeakPolarContact) : 0u);
    kind |= repel ? (1u << ARP_IonicRepulsion) : 0u;
    kind |= hphob ? (1u << ARP_HydrophobicContact) : 0u;
    return clash ? (1u << ARP_StericClash) : kind;                                              // complex.rs:233-235
}

// The same rules for the hot kernels (no probes), table-driven.  Everything that does not depend on the element pair is a
// function of the seven pair predicates W = (Pa & Qb) | (Pb & Qa) (Fat::pw) and the distance level L = number of the nested
// bounds {4.5, 4.0, 3.5} the pair is inside: one 512-entry LDS table read replaces ~30 vector instructions of bit logic.
// Entry = the rows IonicBond / PolarContact / WeakPolarContact / IonicRepulsion / HydrophobicContact (complex.rs:238-296 without a
// probe), bit 30 = "a donor..acceptor pair within 4.0 A" (hbond.rs:37,81: a hydrogen probe decides if a residue carries hydrogens),
// bit 29 = CYS SG pair (vdw.rs:46-53: the dihedral probe decides inside the covalent band).
__host__ __device__ constexpr inline uint32_t pair_lut_entry(uint32_t idx) {
    const uint32_t W = idx & 0x7Fu, L = idx >> 7;
    const uint32_t strong = W & 1u, weak = (W >> 1) & 1u, ion = (W >> 2) & 1u, rep = ((W >> 3) | (W >> 4)) & 1u, hy = (W >> 5) & 1u, sg = (W >> 6) & 1u;
    const uint32_t near45 = L >= 1u, near4 = L >= 2u, near35 = L >= 3u;
    const uint32_t ionic = near4 & ion;                                                          // ionic.rs:11-22,37-57
    uint32_t k = ionic << ARP_IonicBond;
    k |= (strong & near35 & (ionic ^ 1u)) << ARP_PolarContact;                                   // complex.rs:240-251 without a probe
    k |= (weak & near35) << ARP_WeakPolarContact;
    k |= (near4 & rep) << ARP_IonicRepulsion;                                                    // ionic.rs:25-35,59-81
    k |= (near45 & hy) << ARP_HydrophobicContact;                                                // hydrophobic.rs:10-24
    k |= (near4 & (strong | weak)) << 30;
    k |= sg << 29;
    return k;
}
// A pair is handed to the deferred pass (which runs classify<true>, exact for every pair) when a probe MAY be needed: a
// donor/acceptor pair within 4.0 A where either residue carries hydrogens (a superset of hbond.rs:37-42,81-86: the reference
// only looks at the donor's residue), or a CYS SG..SG pair in the covalent band (vdw.rs:46-53).
DEVFN uint32_t classify_fast(const LdsParams &prm, double s, uint32_t pa, uint32_t pb, uint32_t have_res) {
    const uint32_t e = ((pa & ARP_ATTR_ELEM_MASK) << 4) | (pb & ARP_ATTR_ELEM_MASK);
    const bool clash = s < prm.s_clash[e], cov = s < prm.s_cov[e], vdw = s < prm.s_vdw[e];  // vdw.rs:32-43 (strict <)
    const uint32_t L = (s < prm.s_hphob ? 1u : 0u) + (s < prm.s_ion ? 1u : 0u) + (s < prm.s_polar ? 1u : 0u);  // d <= 4.5 / 4.0 / 3.5
    const uint32_t W = (((pa >> 8) & (pb >> 16)) | ((pb >> 8) & (pa >> 16))) & 0x7Fu;
    const uint32_t t = prm.lut[W | (L << 7)];
    const uint32_t probe = ((t & (pa | pb)) >> 30) | ((cov ? 1u : 0u) & (t >> 29) & have_res);  // (kPwResHasH = bit 30, like the table's flag)
    uint32_t kind = (t & 0x1FFFFFFFu) | (cov ? (1u << ARP_CovalentBond) : (vdw ? (1u << ARP_VanDerWaalsContact) : 0u));
    kind = (probe & 1u) ? kDeferKind : kind;
    return clash ? (1u << ARP_StericClash) : kind;                                               // complex.rs:233-235
}

// (float) of the correctly rounded f64 square root -- what the reference stores in the table (mod.rs:148) -- without the
// library sqrt on the common path.  f32 rsq seed r = (1 + e) / sqrt(s), |e| < 2^-21.5; y0 = s r; one residual correction
// y = y0 + (s - y0^2) (r / 2) = sqrt(s) (1 - 1.5 e^2 - ...): relative error < 1.7e-13, i.e. < 1600 ulp(f64).  That cannot change
// the f32 rounding unless y sits within 2048 f64 ulps of an f32 rounding boundary (probability 4e-6 per pair); only then, and
// for degenerate s, the exact library routine runs -- behind a WAVE-UNIFORM branch: left to the compiler, both paths were
// evaluated for every pair.
DEVFN float dist_f32(double s) {
    const double r = (double)__frsqrt_rn((float)s);
    const double y0 = s * r, hr = 0.5 * r;
    const double y = __fma_rn(__fma_rn(-y0, y0, s), hr, y0);
    const uint32_t low = (uint32_t)__double_as_longlong(y) & 0x1FFFFFFFu;  // the 29 bits a cast to f32 drops
    const bool exact = !(s > 1e-30 && s < 1e30) | (low - (0x10000000u - 2048u) <= 4096u);  // degenerate, or near the midpoint
    float out = (float)y;
    if (__ballot(exact) != 0ull) {
        asm volatile("" ::: "memory");  // keep this a branch (no speculation of the long sequence)
        if (exact) out = (float)sqrt(s);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------- pair search + launch
#include "pairs.inl"
#include "pairs_emit.inl"
#include "batch.inl"
#include "sap.inl"

}  // namespace arp
