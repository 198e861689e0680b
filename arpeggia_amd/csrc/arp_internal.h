// Internal declarations shared by the HIP kernels (kernels.hip) and the host engine (engine.cpp).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#include "../../include/arpeggia_amd.h"
#include "debug_knobs.h"

namespace arp {

// Decision constants in SQUARED-distance space.  The reference compares d = sqrt(s) (correctly rounded f64)
// against thresholds T with `<` (vdw.rs:33-41) or `<=` (everything else).  Because rn(sqrt(.)) is monotone,
// {s : sqrt(s) < T} = {s : s < lt(T)} with lt(T) = min{s : sqrt(s) >= T}; likewise d <= T  <=>  s < le(T).
// The host computes lt/le exactly (engine.cpp: bound_lt / bound_le), so the device never needs a sqrt to decide.
struct DevParams {
    double r2;            // squared search radius, inclusive candidate test d^2 <= r2 (rstar locate_within_distance): r2_call, or less with
                          // ARP_FLAG_CONTACTS_ONLY (no rule reaches further than the largest bound of the elements present) -- set by the grid sizing
    double s_ion;         // le(4.0)   ionic.rs:5, hbond.rs:7
    double s_polar;       // le(3.5)   hbond.rs:8
    double s_hphob;       // le(4.5)   hydrophobic.rs:5
    double s_clash[256];  // lt(cov[a]+cov[b] - c)   vdw.rs:33
    double s_cov[256];    // lt(cov[a]+cov[b] + c)   vdw.rs:34
    double s_vdw[256];    // lt(vdw[a]+vdw[b] + c)   vdw.rs:41
    double s_hacc[16];    // le(h_vdw + vdw[acceptor] + c)   hbond.rs:54,98
    double s_cov_max;     // the largest s_cov[] of the element pairs PRESENT: below it a candidate MAY be inside a covalent / clash band (k_emit's short level
                          // count) -- set by the grid sizing
    float r2f;            // prefilter threshold in f32 (r2 + margin), set by the grid sizing
    uint32_t flags;       // arp_params.flags (ARP_FLAG_CONTACTS_ONLY is read by the pair kernels)
    double r2_call;       // dist_cutoff^2 (complex.rs:191) as the caller gave it.  Everything but r2, r2f and s_cov_max is constant for a given
                          // arp_params: the device copy is uploaded when the parameters change, not per call
    uint32_t strip_force; // diagnostics (arp_debug_set "strip_rows"): rows per y strip of the cell order, a power of two; 0 = chosen by input size
    uint32_t pad_;
};

// Uniform grid, written by the device-side setup kernel (no host round trip).
struct GridParams {
    double ox, oy, oz;    // origin = min corner of the heavy atoms
    double inv_edge;      // 1 / cell edge in y and z, edge >= dist_cutoff * (1 + 1e-6)
    double inv_edge_x;    // kx / edge: cells are kx times finer along x (the fastest-running cell index), so that a lane's slot windows
                          // -- x-runs of 2 kx + 1 cells -- hug the search sphere: the same five windows, up to a third fewer tests
    uint32_t kx;          // x cells per cell edge (1, 2 or 4): a neighbour within the cutoff is at most kx cells away along x
    uint32_t nx, ny, nz;  // cells per axis for ONE model
    uint32_t nzt;         // total z layers = n_models * (nz + 1): each model gets its own slab + an empty separator
    uint32_t ncells;      // nx * ny * nzt (ny rounded up to whole strips when sy_shift != 0)
    uint32_t n_heavy;     // atoms in the grid (non-H)
    uint32_t n_tasks;     // ceil(n_heavy / 64): one wave-task per 64 consecutive slots
    uint32_t bad;         // bit 0: non-finite coordinate seen, bit 1: model ids too sparse for the workspace
    float prefilter_margin;
    uint32_t all_both;    // every heavy atom is in the ligand AND the receptor set (groups "/"): orient() takes its short form
    double mx, my, mz;    // box midpoint: the f32 prefilter records are relative to it (halves their magnitude)
    double r2m;           // prefilter threshold on d^2 in f64: r2 + storage margin + dot-form margin
    const double *model_org;  // packed batches: per model {origin xyz, midpoint xyz} -- every member sits in a grid slab of its own
                              // position, however far apart the members are in space; nullptr = one origin for all models
    uint32_t rk_bad;      // k_place met a residue ordinal or chain rank that does not fit the 32-bit residue word (Sorted::rkey): the residue-rule
                          // kernels (k_emit<.., RES>) then reject nothing early -- the exact phase decides on the real keys, as always
    uint32_t sy_shift;    // cell rows run (layer, y) when 0; else in y strips of 2^sy_shift rows: (strip, layer, y inside the strip) -- grid_row
};

// Row of cells (y, layer) -> its place in the cell order; a row's nx cells are consecutive.  Layer-major order keeps a row's neighbours in the
// next layer a whole layer of atoms away: the emit kernel's gathers find them in the XCD's 4 MB L2 only while two layers of records fit there
// (up to ~2 x 10^6 atoms of a compact structure).  Beyond that the rows are ordered in y strips (grid_setup picks the height), and the distance
// is a strip's share of a layer.  ny: the real row count (sy_shift == 0); strips pad it to a multiple of their height with empty rows.
__host__ __device__ inline uint32_t grid_row(uint32_t y, uint32_t layer, uint32_t ny, uint32_t nzt, uint32_t s) {
    return s ? ((((y >> s) * nzt + layer) << s) | (y & ((1u << s) - 1u))) : layer * ny + y;
}
__host__ __device__ inline void grid_row_decode(uint32_t r, uint32_t ny, uint32_t nzt, uint32_t s, uint32_t &y, uint32_t &layer) {
    if (s) { const uint32_t q = r >> s; layer = q % nzt; y = ((q / nzt) << s) | (r & ((1u << s) - 1u)); }
    else { y = r % ny; layer = r / ny; }
}

// Device view of the caller's SoA (all device pointers).
struct DevAtoms {
    uint32_t n;
    const double *x, *y, *z;
    const uint32_t *attr, *res_ord;
    const uint32_t *chain_rank, *model;
    const uint32_t *res_id, *res_h_ptr, *res_h_idx, *res_cb, *res_sg;
    uint32_t n_res;
    uint32_t per_model;   // packed batch: size the grid by the largest member and give every model its own origin
};

// Exact-phase record of one heavy atom, 48 B = three 16-byte parts.  The hot kernels read parts 0 and 1 whole and the first
// half of part 2; `attr` (the caller's word + the residue-has-hydrogens bit) is only read by the deferred probe passes.
//   pw = pair word, built when the atom is placed (grid.inl make_pair_word): element class in bits 0-3 and again in bits 4-7, the class bits the
//   pair rules combine as two bytes P (bits 8-14) and Q (bits 16-22) such that (Pa & Qb) | (Pb & Qa) has one bit per pair
//   predicate, LIGAND / RECEPTOR in bits 24 / 25, residue-has-hydrogens in bit 30.
//   {res_ord, crm} sit side by side: read as one 64-bit word they are the key (model, chain rank, residue ordinal) that orders the two atoms
//   of a pair when every atom is in both chain sets (orient_all_both).
struct __attribute__((aligned(16))) Fat {
    double x, y;
    double z; uint32_t pw, orig /* index into the caller's arrays */;
    uint32_t res_ord, crm /* chain rank (all 32 bits; the model is not in the key: every model owns its own slab of the grid + an empty
                             separator layer, so the windows of an atom only ever hold atoms of its own model, complex.rs:96-98) */, cell /* cell id of the slot */, attr;
};
constexpr uint32_t kPwLigand = 1u << 24, kPwReceptor = 1u << 25, kPwResHasH = 1u << 30;  // (bits 26-29 and 31 stay clear: the emit kernel ANDs the word with a table entry's probe bits)

// Cell-sorted copy of the heavy atoms (slot order: x-major cells, atoms of a cell in ascending input index).
struct Sorted {
    float4 *rec;          // {x-mx, y-my, z-mz as f32, |.|^2 of those three}  -- prefilter operand, 16 B
    Fat *fat;
    uint32_t *rkey;       // residue word chain rank << 20 | residue ordinal (kRkOrdBits): two atoms whose words differ by at most 1 are of one residue
                          // or of sequence neighbours in one chain and never pair (complex.rs:108-113) -- k_emit<.., RES> drops such prefilter
                          // survivors before the gathers.  Written by k_place<true> only, valid while GridParams::rk_bad == 0
};
// the residue word: 20 bits of residue ordinal (at most 2^20 - 3, so that the words of two chains are at least 3 apart) + 11 bits of chain rank
// (bit 31 stays clear: the home side disables the rule with a word no neighbour word comes within 2^30 of)
constexpr uint32_t kRkOrdBits = 20, kRkOrdMax = (1u << kRkOrdBits) - 3u, kRkChainMax = (1u << 11) - 1u, kRkOff = 0xC0000000u;

// Same-address (and same-line) device atomics serialise in one L2 channel at ~90 ns each: the task counters of the eight XCD
// groups sit in separate 128-byte lines (measured: sharing one line cost the emit kernel 155 us of hand-out time).
constexpr uint32_t kTaskCtrStride = 32;                       // words
constexpr uint32_t kTaskCtrWords = 4 * 8 * kTaskCtrStride;    // [mode][group]

struct Workspace {
    double *partials;         // k_bounds: [256][8] per-block partial results
    uint32_t *tickets;        // self-resetting arrival counters: [0] bounds, [1] cell scan, [2] pair scan
    GridParams *grid;
    DevParams *params;
    uint32_t *cell_of_atom;   // n
    uint32_t *rank_of_atom;   // n: arrival rank inside the cell (returning atomic)
    uint32_t *cell_count;     // ncells_cap + 1
    uint32_t *cell_start;     // ncells_cap + 1
    uint32_t *perm;           // n: slot -> atom (arrival order inside a cell)
    uint32_t *slot_cell;      // n
    Sorted sorted;
    uint32_t *task_count;     // n/64 + 2: candidate pairs per wave-task
    unsigned long long *task_base;  // n/64 + 2
    uint32_t *scan_tmp;       // block sums (1024 + 1)
    unsigned long long *scan_tmp64;
    unsigned long long *result;  // [0] = total pairs, [1] = flags, [2] = emit allocator head (64-record units), [3] = deferred candidates,
                                 // [4] = how many of the first 255 atoms carry their predecessor's residue word (k_place: the launcher's hint for the next call)
    ulonglong2 *hole_list;    // emit mode: one (start, length) per block
    arp_pair *scratch;        // emit mode: home of positions >= the caller's capacity until k_fixup has closed the holes
    unsigned long long scratch_cap;
    uint32_t *task_ctr;       // [4 modes][8 block groups] x kTaskCtrStride words: next wave-task of the group, one counter per 128-B line
    uint2 *defer_list;        // emit mode: candidates whose classification needs a hydrogen / disulfide probe
    unsigned long long defer_cap;
    uint32_t ncells_cap;
    uint32_t n_cap;
    uint32_t *model_box;      // per-model bounding boxes of a packed batch: 65536 x {min xyz, max xyz} as order-preserving f32 codes
    double *model_org;        // 65536 x {origin xyz, midpoint xyz}
};

// Device view of one pack (batch.inl): the members' arrays back to back + the descriptor table {first atom, first residue, first
// hydrogen-list entry, model offset} x (K + 1) and the per-member scratch of the renumbering and of the pair-list split.
struct PackDesc {             // entry m of K + 1 (the last one is the sentinel holding the totals)
    uint32_t first_atom, first_res, first_h, model_off;
};
struct PackArrays {
    uint32_t n, n_res, n_h, K;
    PackDesc *desc;
    uint32_t *model;
    uint32_t *res_id, *res_h_ptr, *res_cb, *res_sg, *res_h_idx;
    uint32_t *n_models, *status;             // K, 1
    unsigned long long *count, *offset, *cursor;  // K, K + 1, K
};

struct Profiler {
    static constexpr int kMax = 32;
    bool enabled = false;
    int n = 0;
    const char *names[kMax];
    hipEvent_t ev0[kMax], ev1[kMax];
    bool created = false;
    void begin(const char *name, hipStream_t st);
    void end(hipStream_t st);
};

// Launch wrappers (kernels.hip / pairs.inl).  All asynchronous on `st`.
void launch_grid(const DevAtoms &in, const Workspace &ws, hipStream_t st, Profiler *prof, double cutoff, bool ordered, bool want_rkey = false);
void launch_count(const DevAtoms &in, const Workspace &ws, hipStream_t st, Profiler *prof, unsigned long long capacity, bool have_out, bool contacts_only);
void launch_fill_ordered(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof,
                         bool contacts_only);
bool launch_emit(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof, bool contacts_only,
                 bool skip_deferred, bool res_filter = false);  // true: the hole-free sequence of small inputs ran (the host derives result[0] and the flags: engine.cpp finish_result)
bool emit_takes_res_filter(const DevAtoms &in);  // the single-pass emitter has residue-rule kernels for an input of this size (the grid build then writes Sorted::rkey)
unsigned long long emit_scratch_records();
void launch_neighbor_sum(const DevAtoms &in, const Workspace &ws, double radius, double r2, const float *weight, float *out, hipStream_t st, Profiler *prof);
void launch_pack_fix(const PackArrays &pa, hipStream_t st);
void launch_pack_split(const PackArrays &pa, const unsigned long long *result, const arp_pair *pairs, unsigned long long capacity, arp_pair *grouped, bool ordered, hipStream_t st);

}  // namespace arp
