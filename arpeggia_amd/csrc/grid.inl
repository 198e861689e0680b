// Uniform-grid (cell list) construction.  Included by kernels.hip inside namespace arp.
//
//   k_bounds   bounding box of the heavy atoms as per-block partial results
//   k_cellid   sizes the grid from those (every block for itself, block 0 publishes: no launch of its own for one block's work), then the cell
//              of every atom + its arrival rank in the cell (one returning atomic per distinct cell of a block).  Inputs of a few thousand
//              atoms skip k_bounds as well: every block of k_cellid reads all the atoms.  Packed batches size their grid per model in k_setup.
//   scan       cell_count -> cell_start: one block while the cells fit its registers, two launches beyond; clears cell_count for the next call
//   k_place    emit mode: each atom writes its records straight to slot cell_start + arrival rank (coalesced reads)
//   k_scatter + k_gather   ordered mode: slots inside a cell follow the atom index, so the emitted order is reproducible

// ---------------------------------------------------------------------------------------------- bounds + grid setup
// (r2 = the call's squared search radius; returns the f32 prefilter threshold of the two-pass kernels, DevParams::r2f)
// y strips (arp_internal.h grid_row) once a layer of cells holds more than kStripLayerAtoms atoms, the largest power of two of rows that keeps a strip's share of
// a layer at or below kStripAtoms.  Measured on one box, strips of 16 rows against layer order (tests/microbench/ab_r5m.sh; S2, emit us): 2 x 10^6 atoms
// (30 k per layer) 296 / 297, 3 x 10^6 (40 k) 504 / 470, 4 x 10^6 (49 k) 687 / 633, 6 x 10^6 (64 k) 1083 / 945, 8 x 10^6 (77 k) 1466 / 1273; strips of 8 and 32
// rows are both slower than 16 at 4 and 8 x 10^6 (ab_r5k.sh): 16 rows there are 9 500 / 12 000 atoms per strip and layer.
constexpr uint32_t kStripLayerAtoms = 36864, kStripAtoms = 12288;
DEVFN float grid_setup(const double lo_in[3], const double hi_in[3], bool empty, uint32_t n_models, uint32_t bad, GridParams *g, double r2,
                       double cutoff, uint32_t ncells_cap, uint32_t n_atoms, uint32_t strip_force) {
    double lo[3], ext[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = empty ? 0.0 : lo_in[k];
        ext[k] = empty ? 0.0 : hi_in[k] - lo[k];
        if (!(ext[k] >= 0.0) || !isfinite(ext[k])) ext[k] = 0.0;
        if (!isfinite(lo[k])) lo[k] = 0.0;
    }
    // every model owns at least two z layers (a slab + its separator): model ids the workspace cannot hold (sparse ids such as
    // 65535 on a ten-atom input) are an input error -- clamped here so that the cell arithmetic stays in range, flagged in `bad`
    uint32_t nm = n_models ? n_models : 1u;
    if (2ull * nm > (unsigned long long)ncells_cap) { nm = ncells_cap / 2u; bad |= 2u; }
    // edge slightly above |cutoff| so that |dx| <= |cutoff| can never straddle two cell boundaries after rounding (the
    // reference only ever uses cutoff^2, complex.rs:191, so a negative cutoff searches the same sphere)
    double edge = fabs(cutoff) * (1.0 + 1e-6);
    if (!(edge > 1e-3)) edge = 1e-3;
    // cells kx = 4, 2 or 1 times finer along x (arp_internal.h GridParams::kx): the finest split that keeps the cell count below about a
    // third of the atom count.  Finer cells cut prefilter tests (S2 10^6 atoms: emit 188 / 177 / 174 us at kx = 1 / 2 / 4) but every cell
    // is an entry of the count / scan / start arrays: a pack of 1250 five-thousand-atom structures, each model with its own slab of cells
    // sized by the largest member, builds its grid in 340 / 393 / 455 us (profiles/r03_kx_sweep.txt).
    double nx, ny, nz;
    uint32_t kx = 1u;
    const double soft_cap = fmin((double)ncells_cap, fmax(0.35 * (double)n_atoms, 4096.0));
    for (int it = 0;; ++it) {
        ny = floor(ext[1] / edge) + 1.0; nz = floor(ext[2] / edge) + 1.0;
        for (kx = 4u; kx > 1u; kx >>= 1) {
            nx = floor(ext[0] * (double)kx / edge) + 1.0;
            if (nx * ny * (nz + 1.0) * (double)nm <= soft_cap) break;
        }
        nx = floor(ext[0] * (double)kx / edge) + 1.0;
        if (nx * ny * (nz + 1.0) * (double)nm <= (double)ncells_cap) break;
        if (it >= 512) { nx = ny = nz = 1.0; kx = 1u; edge = INFINITY; break; }  // unreachable for finite extents (1.26^512 overflows first); a bound, not a hope
        edge *= 1.2599210498948732;  // sparse / huge extents: coarser cells stay correct (edge >= cutoff)
    }
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2];
    g->inv_edge = isfinite(edge) ? 1.0 / edge : 0.0;
    g->inv_edge_x = isfinite(edge) ? (double)kx / edge : 0.0;
    g->kx = kx;
    g->nx = (uint32_t)nx; g->ny = (uint32_t)ny; g->nz = (uint32_t)nz;
    g->nzt = nm * (g->nz + 1u);
    // y strips (arp_internal.h grid_row): one model only -- a pack's members must stay contiguous in the cell order (batch.inl splits by slot range)
    uint32_t sy = 0u;
    if (nm == 1u && g->ny > 2u) {
        const double layer_atoms = (double)n_atoms / nz;
        double rows = strip_force ? (double)strip_force : (layer_atoms > (double)kStripLayerAtoms ? fmax(ny * (double)kStripAtoms / layer_atoms, 8.0) : 0.0);
        while (rows >= 2.0 && (2u << sy) <= g->ny - 1u) { rows *= 0.5; sy++; }  // the largest power of two <= rows that leaves at least two strips
        const unsigned long long padded = (unsigned long long)g->nx * (((g->ny + (1u << sy) - 1u) >> sy) << sy) * g->nzt;
        if (padded > (unsigned long long)ncells_cap) sy = 0u;
    }
    g->sy_shift = sy;
    g->ncells = g->nx * (sy ? (((g->ny + (1u << sy) - 1u) >> sy) << sy) : g->ny) * g->nzt;
    g->n_heavy = 0; g->n_tasks = 0;
    g->bad = bad;
    g->rk_bad = 0u;
    // f32 prefilter: relative coordinates carry <= 2^-24 * extent of rounding each; a 10x-safe bound on the
    // induced error of dx^2+dy^2+dz^2 near the cutoff (derivation in DESIGN.md "Prefilter margin")
    const double M = fmax(ext[0], fmax(ext[1], ext[2])) + edge;
    const double margin = 4e-6 * (r2 + fabs(cutoff) * M) + 1e-6;
    g->prefilter_margin = (float)margin;
    // The test is evaluated as |n|^2 - 2 n.h <= thr - |h|^2 (3 FMAs instead of 3 subtractions + 3 multiply-adds) on records
    // centred on the box midpoint, |coordinate| <= C = M / 2.  Rounding of the stored |n|^2 and of the three FMAs moves the
    // left side by at most 24 * 2^-24 * C^2 (DESIGN.md "Prefilter margin"); twice that is added to the threshold.
    for (int k = 0; k < 3; k++) (&g->mx)[k] = lo[k] + 0.5 * ext[k];
    const double C = 0.5 * M;
    g->r2m = r2 + margin + 3e-6 * C * C;
    return __double2float_ru(r2 + margin);
}

// Bounding box: per-block partial results with plain stores; the blocks of the NEXT kernel reduce them (k_cellid, or k_setup for packs).
// (A single launch with an arrival ticket was measured 2-3x slower: the per-block device-scope atomics / write-through
// stores cost more than the extra launch.)  partials: [kBoundsBlocks][8] doubles = {min xyz, max xyz, models, bad}.
constexpr uint32_t kBoundsBlocks = 256, kBoundsThreads = 1024;
constexpr uint32_t kScanBlocks = 1024, kScanThreads = 256, kScanPartAt = 32;  // (the two-launch scan's block decomposition; k_scan_single's chunk totals live behind the 32 result words)
constexpr uint32_t kScanSingleBlocks = kScanThreads;  // one published word per thread of a later block: a single round of device-scope loads
struct BoxAcc {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t models = 0, bad = 0;
};
struct BoxLds { double mn[kBoundsThreads / 64][3], mx[kBoundsThreads / 64][3]; uint32_t models[kBoundsThreads / 64], bad[kBoundsThreads / 64]; };
// Order-preserving codes of f32 values (unsigned compare = float compare): the box only has to CONTAIN the atoms -- cell indices are clamped, and
// two atoms within the cutoff stay in adjacent cells under any monotone clamp -- so it is reduced as f32 values rounded outwards.
DEVFN uint32_t f32_code(float f) { const uint32_t u = __float_as_uint(f); return (u >> 31) ? ~u : (u | 0x80000000u); }
DEVFN float f32_decode(uint32_t c) { return __uint_as_float((c >> 31) ? (c & 0x7FFFFFFFu) : ~c); }
DEVFN uint32_t wave_or_u32(uint32_t v) {  // (wave_reduce_u32 with |)
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <uint32_t NW>
DEVFN void box_block_reduce(BoxAcc &a, BoxLds &l) {
    // Inside a wave on the vector ALU's data-parallel primitives, as f32 codes (round 5: the f64 butterfly was 26 ds_bpermute round trips per
    // stage, 156 per reduction, each waited for -- most of the grid sizing's time inside k_cellid and a third of k_bounds')
    for (int k = 0; k < 3; k++) {
        a.mn[k] = (double)f32_decode(wave_min_u32(f32_code(__double2float_rd(a.mn[k]))));
        a.mx[k] = (double)f32_decode(wave_max_u32(f32_code(__double2float_ru(a.mx[k]))));
    }
    a.models = wave_max_u32(a.models);
    a.bad = wave_or_u32(a.bad);
    const uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; k++) { l.mn[w][k] = a.mn[k]; l.mx[w][k] = a.mx[k]; }
        l.models[w] = a.models; l.bad[w] = a.bad;
    }
    __syncthreads();
    for (uint32_t v = 0; v < NW; v++) {
        for (int k = 0; k < 3; k++) { a.mn[k] = fmin(a.mn[k], l.mn[v][k]); a.mx[k] = fmax(a.mx[k], l.mx[v][k]); }
        a.models = max(a.models, l.models[v]); a.bad |= l.bad[v];
    }
}

// atoms i0, i0 + stride, ... into the thread's accumulators.  Four independent atoms per trip, loaded UNCONDITIONALLY from a clamped index:
// predicated loads make hipcc wait for each attr word before issuing the next atom's loads, which turns the loop into serial round trips.
DEVFN void box_accumulate(const DevAtoms &in, uint32_t first, uint32_t stride, BoxAcc &acc) {
    const uint32_t last = in.n ? in.n - 1u : 0u;
    for (uint32_t i0 = first; i0 < in.n; i0 += 4u * stride) {
        double p[4][3];
        uint32_t at[4], md[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = min(i0 + (uint32_t)u * stride, last);
            at[u] = in.attr[i]; md[u] = in.model[i];
            p[u][0] = in.x[i]; p[u][1] = in.y[i]; p[u][2] = in.z[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool use = (i0 + (uint32_t)u * stride < in.n) & !(at[u] & ARP_ATTR_H);
            for (int k = 0; k < 3; k++) {
                acc.bad |= (use & !isfinite(p[u][k])) ? 1u : 0u;
                acc.mn[k] = use ? fmin(acc.mn[k], p[u][k]) : acc.mn[k];
                acc.mx[k] = use ? fmax(acc.mx[k], p[u][k]) : acc.mx[k];
            }
            acc.models = use ? max(acc.models, min(md[u], 0xFFFFFFFEu) + 1u) : acc.models;
            acc.bad |= use ? (0x100u << (at[u] & ARP_ATTR_ELEM_MASK)) : 0u;  // bits 8..23: element classes present in the grid
            acc.bad |= (use & ((at[u] & (ARP_ATTR_LIGAND | ARP_ATTR_RECEPTOR)) != (ARP_ATTR_LIGAND | ARP_ATTR_RECEPTOR))) ? (1u << 24) : 0u;  // bit 24: an atom outside L or R
        }
    }
}
DEVFN void box_from_partials(const double *partials, uint32_t n_partials, BoxAcc &acc) {
    for (uint32_t b = threadIdx.x; b < n_partials; b += blockDim.x) {
        const double *p = partials + 8 * b;
        for (int k = 0; k < 3; k++) { acc.mn[k] = fmin(acc.mn[k], p[k]); acc.mx[k] = fmax(acc.mx[k], p[3 + k]); }
        acc.models = max(acc.models, (uint32_t)p[6]); acc.bad |= (uint32_t)p[7];
    }
}

__global__ __launch_bounds__(kBoundsThreads) void k_bounds(DevAtoms in, double *partials) {
    __shared__ BoxLds l;
    BoxAcc acc;
    box_accumulate(in, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, acc);
    box_block_reduce<kBoundsThreads / 64>(acc, l);
    if (threadIdx.x == 0) {
        double *p = partials + 8 * blockIdx.x;
        for (int k = 0; k < 3; k++) { p[k] = acc.mn[k]; p[3 + k] = acc.mx[k]; }
        p[6] = (double)acc.models; p[7] = (double)acc.bad;
    }
}

// Packed batches: bounding box PER MODEL, as order-preserving codes of f32 values rounded outwards (the box only has to contain
// the atoms).  The members of a pack are contiguous, so a wave is almost always inside one model: six wave reductions, six atomics.
constexpr uint32_t kPackModels = 65536;  // entries of Workspace::model_box / model_org: every index into them is clamped to this
__global__ __launch_bounds__(256) void k_model_box_init(uint32_t *box) {  // {min xyz = +inf code, max xyz = -inf code} per model
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    box[i] = (i % 6u) < 3u ? 0xFFFFFFFFu : 0u;
}
__global__ __launch_bounds__(256) void k_model_bounds(DevAtoms in, uint32_t *box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const bool use = i < in.n && !(in.attr[i] & ARP_ATTR_H);
    uint32_t m = ARP_NONE, lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (use) {
        m = min(in.model[i], kPackModels - 1u);  // (k_pack_fix keeps a pack's ordinals below this; the clamp is the table's own guard)
        const double p[3] = {in.x[i], in.y[i], in.z[i]};
        for (int k = 0; k < 3; k++) { lo[k] = f32_code(__double2float_rd(p[k])); hi[k] = f32_code(__double2float_ru(p[k])); }
    }
    const uint32_t first = wave_min_u32(m);
    if (first == ARP_NONE) return;
    if (__all(m == first || m == ARP_NONE)) {
        for (int k = 0; k < 3; k++) { lo[k] = wave_min_u32(lo[k]); hi[k] = wave_max_u32(hi[k]); }
        if (lane == 0) for (int k = 0; k < 3; k++) { atomicMin(&box[6u * first + k], lo[k]); atomicMax(&box[6u * first + 3 + k], hi[k]); }
    } else if (use) {
        for (int k = 0; k < 3; k++) { atomicMin(&box[6u * m + k], lo[k]); atomicMax(&box[6u * m + 3 + k], hi[k]); }
    }
}

// Sizes the grid from a block's box accumulators.  Called by all 256 threads of a block; on return (behind a barrier) l.g holds the grid
// parameters.  Every block of k_cellid does this for itself -- the inputs are a few KB out of the L2 and the work is one thread's -- which
// saves the launch a single-block kernel would cost; the block with `publish` also writes the grid, the call's derived parameters
// (DevParams::r2, r2f, s_cov_max) and the per-call zeroes to global memory for the kernels that follow.  Nothing read here is written here:
// the caller's squared cutoff stays in DevParams::r2_call.
struct SetupLds { BoxLds box; double red[4]; double ext[4][3]; GridParams g; };
DEVFN void setup_block(BoxAcc &acc, SetupLds &l, bool publish, GridParams *g, DevParams *prm, double cutoff, uint32_t ncells_cap, uint32_t n_atoms,
                       unsigned long long *result, uint32_t *task_ctr, const uint32_t *model_box, double *model_org) {
    if (publish) {
        for (uint32_t k = threadIdx.x; k < kTaskCtrWords; k += blockDim.x) task_ctr[k] = 0;  // per-call state of the later kernels
        if (threadIdx.x < 32) result[threadIdx.x] = 0;  // [0..3] the call's results
        // (result + 32 ..: the published chunk totals of k_scan_single, Workspace::result holds kResultWordsAll words)
        for (uint32_t k = threadIdx.x; k < kScanBlocks; k += blockDim.x) result[kScanPartAt + k] = 0ull;
    }
    box_block_reduce<4>(acc, l.box);
    auto block_max = [&](double b) {  // (of non-negative values: the bit patterns of non-negative doubles order like unsigned integers, high word first)
        const uint32_t hi = wave_max_u32((uint32_t)__double2hiint(b));
        const uint32_t lo = wave_max_u32((uint32_t)__double2hiint(b) == hi ? (uint32_t)__double2loint(b) : 0u);
        b = __hiloint2double((int)hi, (int)lo);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) l.red[threadIdx.x >> 6] = b;
        __syncthreads();
        return fmax(fmax(l.red[0], l.red[1]), fmax(l.red[2], l.red[3]));
    };
    const uint32_t present = (acc.bad >> 8) & 0xFFFFu, ea = threadIdx.x >> 4, eb = threadIdx.x & 15u;
    const bool pair_present = (present >> ea) & (present >> eb) & 1u;
    // the largest covalent bound among the element pairs that are PRESENT (k_emit's short level count: a candidate above it is in no
    // covalent or clash band; the parameter table also holds the metals' radii, which would put every other candidate below it)
    const double cov_max = block_max(pair_present ? prm->s_cov[threadIdx.x] : 0.0);
    // ARP_FLAG_CONTACTS_ONLY: no rule can match beyond the largest decision bound of the element pairs that are present
    // (every rule of classify() is `s < bound`), so the search radius shrinks to it -- for C/N/O/S that is the 4.5 A of the
    // hydrophobic rule -- and the dropped candidates are exactly ones the flag would have filtered out.
    double r2 = prm->r2_call;
    if (prm->flags & ARP_FLAG_CONTACTS_ONLY) {
        double b = fmax(prm->s_hphob, fmax(prm->s_ion, prm->s_polar));
        if (pair_present) b = fmax(b, fmax(prm->s_clash[threadIdx.x], fmax(prm->s_cov[threadIdx.x], prm->s_vdw[threadIdx.x])));
        b = block_max(b);
        if (b < r2) { r2 = b; cutoff = sqrt(b); }
    }
    // Packed batch: the grid is sized by the LARGEST member and every model gets its own origin (its box's min corner) and
    // its own midpoint for the f32 records -- members may sit anywhere in space without inflating the cell count.
    double ext[3] = {0.0, 0.0, 0.0};
    const uint32_t box_models = min(acc.models, kPackModels);  // (the per-model tables hold kPackModels entries)
    if (model_box) {  // (differences of f32 values are exact in f64)
        for (uint32_t m = threadIdx.x; m < box_models; m += blockDim.x) {
            const uint32_t *b = model_box + 6u * m;
            if (b[0] == 0xFFFFFFFFu) continue;  // no heavy atom in this model
            for (int k = 0; k < 3; k++) ext[k] = fmax(ext[k], (double)f32_decode(b[3 + k]) - (double)f32_decode(b[k]));
        }
        for (int off = 32; off; off >>= 1) for (int k = 0; k < 3; k++) ext[k] = fmax(ext[k], __shfl_xor(ext[k], off));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; k++) l.ext[threadIdx.x >> 6][k] = ext[k];
        __syncthreads();
        for (int k = 0; k < 3; k++) ext[k] = fmax(fmax(l.ext[0][k], l.ext[1][k]), fmax(l.ext[2][k], l.ext[3][k]));
        if (publish)
            for (uint32_t m = threadIdx.x; m < box_models; m += blockDim.x) {
                const uint32_t *b = model_box + 6u * m;
                double *o = model_org + 6u * m;
                for (int k = 0; k < 3; k++) {
                    const double lo = b[0] == 0xFFFFFFFFu ? 0.0 : (double)f32_decode(b[k]);
                    o[k] = lo; o[3 + k] = lo + 0.5 * ext[k];
                }
            }
    }
    if (threadIdx.x == 0) {
        GridParams gl;
        const double zero[3] = {0.0, 0.0, 0.0};
        const float r2f = grid_setup(model_box ? zero : acc.mn, model_box ? ext : acc.mx, !(acc.mn[0] <= acc.mx[0]), acc.models, acc.bad & 0xFFu, &gl, r2, cutoff, ncells_cap, n_atoms, prm->strip_force);
        gl.model_org = model_box ? model_org : nullptr;
        gl.all_both = (acc.bad >> 24) & 1u ? 0u : 1u;
        l.g = gl;
        if (publish) {
            *g = gl; prm->r2 = r2; prm->r2f = r2f; prm->s_cov_max = cov_max;
            // the input errors as status flags right away (k_fixup and the count scan set them again; the hole-free sequence of small inputs has neither)
            const unsigned long long fl = ((gl.bad & 1u) ? 4ull : 0ull) | ((gl.bad & 2u) ? 64ull : 0ull);
            if (fl) atomicOr(&result[1], fl);
        }
    }
    __syncthreads();
}

// The grid sizing as a kernel of its own: packed batches (the per-model boxes have to be complete first) and empty inputs.
__global__ __launch_bounds__(256) void k_setup(const double *partials, uint32_t n_partials, GridParams *g, DevParams *prm, double cutoff,
                                               uint32_t ncells_cap, uint32_t n_atoms, unsigned long long *result, uint32_t *task_ctr, const uint32_t *model_box, double *model_org) {
    __shared__ SetupLds l;
    BoxAcc acc;
    box_from_partials(partials, n_partials, acc);
    setup_block(acc, l, true, g, prm, cutoff, ncells_cap, n_atoms, result, task_ctr, model_box, model_org);
}

DEVFN uint32_t cell_index(const GridParams &g, double x, double y, double z, uint32_t model) {
    double ox = g.ox, oy = g.oy, oz = g.oz;
    if (g.model_org) { const double *o = g.model_org + 6u * min(model, kPackModels - 1u); ox = o[0]; oy = o[1]; oz = o[2]; }  // packed batch: the member's own corner
    double fx = (x - ox) * g.inv_edge_x, fy = (y - oy) * g.inv_edge, fz = (z - oz) * g.inv_edge;
    uint32_t cx = (fx >= 0.0) ? (uint32_t)fmin(fx, 4.0e9) : 0u;  // NaN -> 0
    uint32_t cy = (fy >= 0.0) ? (uint32_t)fmin(fy, 4.0e9) : 0u;
    uint32_t cz = (fz >= 0.0) ? (uint32_t)fmin(fz, 4.0e9) : 0u;
    cx = min(cx, g.nx - 1u); cy = min(cy, g.ny - 1u); cz = min(cz, g.nz - 1u);
    uint32_t layer = model * (g.nz + 1u) + cz;  // every model owns a z slab followed by one empty layer
    layer = min(layer, g.nzt - 1u);
    return grid_row(cy, layer, g.ny, g.nzt, g.sy_shift) * g.nx + cx;
}

// Cell of every atom and its arrival rank inside the cell.  A block takes 1024 consecutive atoms and merges those that fall into
// the same cell in an LDS hash table first (atoms that are close in the input are close in space: a residue, a chain segment,
// a lattice column), so each distinct cell of the block costs ONE returning device atomic -- scattered device atomics are what
// bounds this kernel (about 14 per ns over the whole chip).  rank = the cell's base for this block + the arrival rank in the block.
constexpr uint32_t kCidThreads = 256, kCidPer = 4, kCidTable = 2048;
// BOX: where the grid parameters come from.  0 = *gp, written by k_setup (packed batches); 1 = this block sizes the grid from k_bounds'
// partial boxes; 2 = from all the atoms, read by this block itself (inputs of a few thousand atoms: a handful of blocks, each a few trips
// over arrays that sit in the L2 -- two launches less on a call that is a chain of launches and little else).  Block 0 publishes.
constexpr uint32_t kCidAllAtoms = 12288;  // BOX = 2 up to here (launch_grid).  Measured against 2048 and "never" (tests/microbench/ab_r4cid.sh): per call
                                          // 44 / 44 / 50 us at 700 atoms, 48 / 52 / 53 at 4000, 53 / 55 / 55 at 8000, 56 / 57 / 57 at 12 000; a 1024-thread
                                          // block for the pass over all the atoms (three trips instead of twelve) loses more to its barriers than it gains
// (BOX != 0 holds the grid sizing's registers: 4 waves per SIMD = 1024 blocks resident, which is why the launcher folds only up to kCidFoldAtoms;
// a pack of 6 x 10^6 atoms ran its 6100 blocks 35 % slower at that occupancy)
constexpr uint32_t kCidFoldAtoms = 1024u * 1024u;
template <int BOX>
__global__ __launch_bounds__(kCidThreads, BOX == 0 ? 8 : 4) void k_cellid(DevAtoms in, GridParams *gp, DevParams *prm, const double *partials, uint32_t n_partials, double cutoff,
                                                        uint32_t ncells_cap, unsigned long long *result, uint32_t *task_ctr, uint32_t *cell_of_atom,
                                                        uint32_t *rank_of_atom, uint32_t *cell_count) {
    __shared__ uint32_t t_key[kCidTable], t_cnt[kCidTable];
    __shared__ SetupLds sl;
    for (uint32_t k = threadIdx.x; k < kCidTable; k += kCidThreads) { t_key[k] = ARP_NONE; t_cnt[k] = 0u; }
    const uint32_t i0 = blockIdx.x * (kCidThreads * kCidPer) + threadIdx.x, last = in.n - 1u;  // (never launched with n == 0)
    double p[kCidPer][3];
    uint32_t at[kCidPer], md[kCidPer];
#pragma unroll
    for (uint32_t u = 0; u < kCidPer; u++) {  // unconditional loads from a clamped index: all of them in flight together (and during the grid sizing)
        const uint32_t i = min(i0 + u * kCidThreads, last);
        at[u] = in.attr[i]; md[u] = (uint32_t)in.model[i];
        p[u][0] = in.x[i]; p[u][1] = in.y[i]; p[u][2] = in.z[i];
    }
    if (BOX != 0) {
        BoxAcc acc;
        if (BOX == 1) box_from_partials(partials, n_partials, acc);
        else box_accumulate(in, threadIdx.x, kCidThreads, acc);
        setup_block(acc, sl, blockIdx.x == 0u, gp, prm, cutoff, ncells_cap, in.n, result, task_ctr, nullptr, nullptr);
    }
    const GridParams g = BOX != 0 ? sl.g : *gp;
    __syncthreads();
    uint32_t c[kCidPer], slot[kCidPer], r[kCidPer];
#pragma unroll
    for (uint32_t u = 0; u < kCidPer; u++) {
        const bool use = (i0 + u * kCidThreads < in.n) & !(at[u] & ARP_ATTR_H);
        c[u] = use ? cell_index(g, p[u][0], p[u][1], p[u][2], md[u]) : ARP_NONE;
        slot[u] = 0u; r[u] = 0u;
        if (use) {
            uint32_t h = (c[u] * 0x9E3779B1u) >> 21;  // 11 bits
            for (;;) {  // linear probing; at most 1024 keys in 2048 entries
                const uint32_t old = atomicCAS(&t_key[h], ARP_NONE, c[u]);
                if (old == ARP_NONE || old == c[u]) break;
                h = (h + 1u) & (kCidTable - 1u);
            }
            slot[u] = h;
            r[u] = atomicAdd(&t_cnt[h], 1u);
        }
    }
    __syncthreads();
    {   // count -> base of this block's atoms in the cell: one returning device atomic per distinct cell of the block.  All of a thread's
        // atomics are issued before the first answer is waited for (round 5: written as one loop, each iteration stored its answer at once and
        // the eight round trips of a thread ran one after the other -- ~8 us of a 19 us kernel whose blocks all run at the same time)
        constexpr uint32_t kPer = kCidTable / kCidThreads;
        uint32_t key[kPer], base[kPer];
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t k = threadIdx.x + j * kCidThreads;
            key[j] = t_key[k];
            base[j] = t_cnt[k];
        }
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++)
            if (key[j] != ARP_NONE) base[j] = atomicAdd(&cell_count[key[j]], base[j]);
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++)
            if (key[j] != ARP_NONE) t_cnt[threadIdx.x + j * kCidThreads] = base[j];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < kCidPer; u++) {
        const uint32_t i = i0 + u * kCidThreads;
        if (i < in.n) { cell_of_atom[i] = c[u]; rank_of_atom[i] = c[u] != ARP_NONE ? t_cnt[slot[u]] + r[u] : 0u; }
    }
}

// ---------------------------------------------------------------------------------------------- scan
// Exclusive scan of in[0..n) (n read from device memory) into out[0..n], out[n] = total, over a fixed 1024-block
// decomposition so that no host knowledge of n is needed.  Three launches: per-block sums, their scan, the per-block
// scans (a fused arrival-ticket variant was measured slower).  ZERO_IN clears the input behind itself (cell_count is
// ready for the next call); FINISH also publishes the pair total and the status flags.
template <typename TOut>
DEVFN TOut block_exclusive_scan(TOut v, TOut *total, TOut *lds /* [kScanThreads/64 + 1] */) {
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
    const uint32_t n = *n_ptr;
    const uint32_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
    const uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
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

template <typename TOut, bool ZERO_IN, bool FINISH>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(uint32_t *in, const uint32_t *n_ptr, const TOut *tmp, TOut *out, const GridParams *g,
                                                             unsigned long long *result, unsigned long long capacity, int have_out) {
    __shared__ TOut lds[kScanThreads / 64 + 1];
    const uint32_t n = *n_ptr;
    const uint32_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
    const uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
    // carry = sum of the partial totals of the blocks before this one, recomputed here from the kScanBlocks raw totals
    // (4-8 KB out of L2): one launch less than scanning them in a kernel of their own
    TOut part = 0, all = 0;
    for (uint32_t k = threadIdx.x; k < kScanBlocks; k += kScanThreads) { const TOut v = tmp[k]; all += v; if (k < blockIdx.x) part += v; }
    TOut carry, grand = 0;
    block_exclusive_scan<TOut>(part, &carry, lds);
    __syncthreads();
    if (blockIdx.x == 0) { block_exclusive_scan<TOut>(all, &grand, lds); __syncthreads(); }
    for (uint32_t base = lo; base < hi; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        TOut v = (i < hi) ? (TOut)in[i] : (TOut)0, tot;
        TOut ex = block_exclusive_scan<TOut>(v, &tot, lds);
        if (i < hi) { out[i] = carry + ex; if (ZERO_IN) in[i] = 0; }
        carry += tot;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const TOut total = grand;
        out[n] = total;
        if (FINISH) {  // the pair-count scan also publishes the result word and the status flags
            result[0] = (unsigned long long)total;
            if (have_out && (unsigned long long)total > capacity) result[1] |= 1ull;
            if (g->bad & 1u) result[1] |= 4ull;
            if (g->bad & 2u) result[1] |= 64ull;
        }
    }
}

// The cell scan of larger inputs in ONE launch (round 5; two before: per-block totals, then carry + apply): every block adds up its chunk and
// publishes the total as one 64-bit word {1, total}; a block's carry is the sum of the words of the blocks before it, read with device-scope
// loads as they appear (the words are zeroed by the grid sizing of the same call, a kernel earlier on the stream).  A block only ever waits
// for blocks with a smaller index -- dispatched before it -- so the wait ends whatever else shares the device; value and flag travel in
// one atomic word, so nothing else has to become visible with it.
__global__ __launch_bounds__(kScanThreads) void k_scan_single(uint32_t *in, const uint32_t *n_ptr, unsigned long long *part, uint32_t *out) {
    __shared__ uint32_t lds[kScanThreads / 64 + 1];
    const uint32_t n = *n_ptr;
    const uint32_t chunk = (n + gridDim.x - 1) / gridDim.x;  // (gridDim.x: kScanSingleBlocks, or kScanBlocks for the cell counts of packs and of inputs beyond 2^20 atoms)
    const uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
    uint32_t s = 0, own;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kScanThreads) s += in[i];
    block_exclusive_scan<uint32_t>(s, &own, lds);
    if (threadIdx.x == 0) __hip_atomic_store(&part[blockIdx.x], (1ull << 32) | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // thread t waits for the words of blocks t, t + 256, ... (blocks with a smaller index were dispatched earlier); the loads of a round are in
    // flight together, and only the words that had not appeared yet are asked for again
    constexpr uint32_t kWordsPer = kScanBlocks / kScanThreads;
    unsigned long long w[kWordsPer];
    bool pending = false;
#pragma unroll
    for (uint32_t j = 0; j < kWordsPer; j++) {
        const uint32_t k = threadIdx.x + j * kScanThreads;
        w[j] = k < blockIdx.x ? __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 32);
    }
#pragma unroll
    for (uint32_t j = 0; j < kWordsPer; j++) pending |= (w[j] >> 32) == 0ull;
    while (pending) {
        pending = false;
#pragma unroll
        for (uint32_t j = 0; j < kWordsPer; j++)
            if ((w[j] >> 32) == 0ull) { w[j] = __hip_atomic_load(&part[threadIdx.x + j * kScanThreads], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pending |= (w[j] >> 32) == 0ull; }
    }
    uint32_t before = 0, carry;
#pragma unroll
    for (uint32_t j = 0; j < kWordsPer; j++) before += (uint32_t)w[j];
    __syncthreads();
    block_exclusive_scan<uint32_t>(before, &carry, lds);
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) out[n] = carry + own;  // (the last block's chunk ends at n: the grand total)
    for (uint32_t base = lo; base < hi; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = (i < hi) ? in[i] : 0u, tot;
        const uint32_t ex = block_exclusive_scan<uint32_t>(v, &tot, lds);
        if (i < hi) { out[i] = carry + ex; in[i] = 0; }  // (clears cell_count behind itself: ready for the next call)
        carry += tot;
        __syncthreads();
    }
}

// The cell scan in ONE launch for inputs whose cells fit one workgroup's registers (launch_grid: up to kScanOneAtoms atoms, i.e. <= 65536
// cells at the 0.35 cells per atom the grid sizing allows): wave w owns 4096 consecutive cells per pass -- 16 coalesced 16-byte loads per
// lane, all in flight together --, scans them with the DPP row shifts (no LDS round trips), and one barrier exchanges the 16 wave totals.
// A sparse input with more cells than that takes further passes (correct, just slower than the two-launch scan the launcher would have
// picked had it known: the cell count only exists on the device).
constexpr uint32_t kScanOneThreads = 1024, kScanOnePer = 16, kScanOneCells = kScanOneThreads * kScanOnePer * 4u, kScanOneAtoms = 180000;
DEVFN uint32_t wave_inclusive_add_u32(uint32_t v) {  // (the scan wave_reduce_u32 is the last lane of)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// Round 5: launched with kScanOneBlocks blocks.  Block b takes the passes b, b + kScanOneBlocks, ...: the usual input is one pass,
// which block 0 runs exactly as before while the other blocks read the cell count and leave; a sparse or many-model input of the same atom
// count (up to 8 n + 64 K cells: ~23 passes at 180 000 atoms) no longer runs them one after the other on one CU.  A pass publishes its total
// as one word {1, total} (the words k_scan_single uses, zeroed by the grid sizing) and adds up the words of the passes before it.
constexpr uint32_t kScanOneBlocks = 32;  // >= ceil((8 * kScanOneAtoms + 65536 + 1) / kScanOneCells) = 23 passes
template <bool ZERO_IN>
__global__ __launch_bounds__(kScanOneThreads) void k_scan_one(uint32_t *in, const uint32_t *n_ptr, unsigned long long *part, uint32_t *out) {
    __shared__ uint32_t wave_total[kScanOneThreads / 64];
    __shared__ uint32_t s_carry;
    const uint32_t n = *n_ptr, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_pass = (n + kScanOneCells - 1u) / kScanOneCells;
    if (blockIdx.x == 0u && n_pass == 0u && threadIdx.x == 0u) out[0] = 0u;  // (no cells: the total)
    for (uint32_t ps = blockIdx.x; ps < n_pass; ps += gridDim.x) {
        const uint32_t base = ps * kScanOneCells;
        const uint32_t w0 = base + wave * (kScanOnePer * 256u) + lane * 4u;
        uint4 v[kScanOnePer];
#pragma unroll
        for (uint32_t j = 0; j < kScanOnePer; j++) {
            const uint32_t i = w0 + j * 256u;
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            if (i + 4u <= n) v[j] = *reinterpret_cast<const uint4 *>(in + i);
            else if (i < n) { v[j].x = in[i]; if (i + 1u < n) v[j].y = in[i + 1u]; if (i + 2u < n) v[j].z = in[i + 2u]; }
        }
        uint32_t ex[kScanOnePer], run = 0;  // ex[j]: cells before this lane's four of trip j, inside the wave's stretch
#pragma unroll
        for (uint32_t j = 0; j < kScanOnePer; j++) {
            const uint32_t s = v[j].x + v[j].y + v[j].z + v[j].w, inc = wave_inclusive_add_u32(s);
            ex[j] = run + inc - s;
            run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        __syncthreads();  // (the previous pass of this block is done with wave_total / s_carry)
        if (lane == 0u) wave_total[wave] = run;
        __syncthreads();
        uint32_t before = 0, pass = 0;
        for (uint32_t k = 0; k < kScanOneThreads / 64u; k++) { const uint32_t t = wave_total[k]; if (k < wave) before += t; pass += t; }
        // the passes before this one (none for the usual single pass): their published totals, read as they appear
        if (n_pass > 1u) {
            if (threadIdx.x == 0u) __hip_atomic_store(&part[ps], (1ull << 32) | pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wave == 0u) {
                uint32_t c = 0;
                for (uint32_t k = lane; k < ps; k += 64u) {
                    unsigned long long w;
                    do { w = __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w >> 32) == 0ull);
                    c += (uint32_t)w;
                }
                c = wave_inclusive_add_u32(c);
                if (lane == 63u) s_carry = c;
            }
            __syncthreads();
            before += s_carry;
        }
#pragma unroll
        for (uint32_t j = 0; j < kScanOnePer; j++) {
            const uint32_t i = w0 + j * 256u, o = before + ex[j];
            const uint4 r = make_uint4(o, o + v[j].x, o + v[j].x + v[j].y, o + v[j].x + v[j].y + v[j].z);
            if (i + 4u <= n) {
                *reinterpret_cast<uint4 *>(out + i) = r;
                if (ZERO_IN) *reinterpret_cast<uint4 *>(in + i) = make_uint4(0u, 0u, 0u, 0u);
            } else if (i < n) {
                out[i] = r.x; if (i + 1u < n) out[i + 1u] = r.y; if (i + 2u < n) out[i + 2u] = r.z;
                if (ZERO_IN) { in[i] = 0u; if (i + 1u < n) in[i + 1u] = 0u; if (i + 2u < n) in[i + 2u] = 0u; }
            }
        }
        if (ps + 1u == n_pass && threadIdx.x == 0u) out[n] = (n_pass > 1u ? s_carry : 0u) + pass;  // the last pass holds the grand total
    }
}

// ---------------------------------------------------------------------------------------------- sort into cells
constexpr uint32_t kAttrResHasH = 0x80000000u;  // internal: the atom's residue carries hydrogens (set when the atom is placed)

// The pair word of an atom (arp_internal.h Fat::pw).  P = {donor, weak donor, POS, POS, NEG, hydrophobic, CYS SG},
// Q = {acceptor, acceptor, NEG, POS, NEG, hydrophobic, CYS SG}: (Pa & Qb) | (Pb & Qa) = {donor..acceptor either way
// (hbond.rs:113-134), weak donor..acceptor (hbond.rs:181-201), POS..NEG (ionic.rs:37-57), POS..POS, NEG..NEG (ionic.rs:59-81),
// hydrophobic pair (hydrophobic.rs:10-24), CYS SG pair (vdw.rs:46-53)}.
DEVFN uint32_t make_pair_word(uint32_t attr, bool res_has_h) {
    const uint32_t don = (attr >> 4) & 1u, acc = (attr >> 5) & 1u, wdon = (attr >> 6) & 1u, pos = (attr >> 7) & 1u, neg = (attr >> 8) & 1u,
                   hyd = (attr >> 9) & 1u, sg = (attr >> 10) & 1u;
    const uint32_t P = don | (wdon << 1) | (pos << 2) | (pos << 3) | (neg << 4) | (hyd << 5) | (sg << 6);
    const uint32_t Q = acc | (acc << 1) | (neg << 2) | (pos << 3) | (neg << 4) | (hyd << 5) | (sg << 6);
    return (attr & ARP_ATTR_ELEM_MASK) | ((attr & ARP_ATTR_ELEM_MASK) << 4) | (P << 8) | (Q << 16) | ((attr & ARP_ATTR_LIGAND) ? kPwLigand : 0u) | ((attr & ARP_ATTR_RECEPTOR) ? kPwReceptor : 0u) |
           (res_has_h ? kPwResHasH : 0u);
}

DEVFN void place_atom(const DevAtoms &in, const GridParams *gp, const Sorted &so, uint32_t i, uint32_t c, uint32_t d) {
    const double x = in.x[i], y = in.y[i], z = in.z[i];
    double mx = gp->mx, my = gp->my, mz = gp->mz;
    if (gp->model_org) { const double *o = gp->model_org + 6u * min((uint32_t)in.model[i], kPackModels - 1u) + 3u; mx = o[0]; my = o[1]; mz = o[2]; }
    const float fx = (float)(x - mx), fy = (float)(y - my), fz = (float)(z - mz);
    const float4 rv = make_float4(fx, fy, fz, (float)((double)fx * fx + (double)fy * fy + (double)fz * fz));
    so.rec[d] = rv;
    // "the residue carries hydrogens" as a bit of the record: the hot kernel never touches the hydrogen tables, the deferred
    // pass resolves residue -> hydrogens itself (hbond.rs:38-42)
    bool has_h = false;
    if (in.n_res) {
        const uint32_t r = in.res_id[i];
        has_h = in.res_h_ptr[r] < in.res_h_ptr[r + 1];
    }
    const uint32_t attr = in.attr[i] & ~kAttrResHasH;
    Fat f;
    f.x = x; f.y = y; f.z = z;
    f.pw = make_pair_word(attr, has_h); f.res_ord = in.res_ord[i];
    f.crm = in.chain_rank[i]; f.orig = i; f.cell = c;
    f.attr = attr | (has_h ? kAttrResHasH : 0u);
    so.fat[d] = f;
}

// Emit mode: slot = cell_start + arrival rank.  Reads are coalesced (input order), each atom writes its 64 bytes once.
// The hydrogen ranges are only looked at when the input carries hydrogens at all (res_h_ptr[n_res] != 0: one scalar load).
// (Round 4 measured four atoms per thread, every independent load in flight before the first dependent one: 30.8 -> 31 us on 10^6 atoms -- the
// counters say the kernel waits on vector-memory ISSUE, i.e. on the scattered 64-byte writes, not on load latency; one atom per thread stays,
// which also keeps four times as many blocks for inputs that do not fill the chip.)
constexpr uint32_t kPlacePer = 1;
// The residue word of an atom (arp_internal.h Sorted::rkey); `bad`: the atom's ordinal or chain rank does not fit it.
DEVFN uint32_t residue_word(uint32_t res_ord, uint32_t chain_rank, bool *bad) {
    *bad = (res_ord > kRkOrdMax) | (chain_rank > kRkChainMax);
    return ((chain_rank & kRkChainMax) << kRkOrdBits) + (res_ord & ((1u << kRkOrdBits) - 1u));
}
// RKEY: also write the residue words of the slots (the launcher is about to run the residue-rule kernels, k_emit<.., RES>).
// Block 0 always leaves result[4] = how many of atoms 1..255 carry their predecessor's residue word: inputs whose residues are runs of atoms
// (every protein; not a cloud of one-atom residues) are the ones the residue-rule kernels pay for, and the engine picks the kernels of the
// NEXT call by it (engine.cpp res_filter_for) -- a choice between two kernels with identical results, never a correctness assumption.
template <bool RKEY>
__global__ __launch_bounds__(256) void k_place(DevAtoms in, GridParams *gp, const uint32_t *cell_start, const uint32_t *cell_of_atom,
                                               const uint32_t *rank_of_atom, Sorted so, unsigned long long *result) {
    const uint32_t i0 = blockIdx.x * (256u * kPlacePer) + threadIdx.x;
    if (i0 == 0) { const uint32_t n_heavy = cell_start[gp->ncells]; gp->n_heavy = n_heavy; gp->n_tasks = (n_heavy + 63u) / 64u; }
    if (in.n == 0u) return;
    const uint32_t last = in.n - 1u;
    if (blockIdx.x == 0u) {  // (block-uniform)
        bool b0, b1;
        const uint32_t ip = threadIdx.x ? threadIdx.x - 1u : 0u, iq = min(threadIdx.x, last);
        const uint32_t wp = residue_word(in.res_ord[min(ip, last)], in.chain_rank[min(ip, last)], &b0), wq = residue_word(in.res_ord[iq], in.chain_rank[iq], &b1);
        const int runs = __syncthreads_count(threadIdx.x != 0u && threadIdx.x <= last && wp == wq && !b0 && !b1);
        if (threadIdx.x == 0u) result[4] = (unsigned long long)runs;
    }
    const bool any_h = in.n_res != 0u && in.res_h_ptr[in.n_res] != 0u;  // (wave-uniform)
    const double *morg = gp->model_org;
    const double gmx = gp->mx, gmy = gp->my, gmz = gp->mz;
    uint32_t c[kPlacePer], rk[kPlacePer], at[kPlacePer], ro[kPlacePer], md[kPlacePer], cr[kPlacePer], rid[kPlacePer];
    double x[kPlacePer], y[kPlacePer], z[kPlacePer];
#pragma unroll
    for (uint32_t u = 0; u < kPlacePer; u++) {  // unconditional loads from a clamped index: all in flight together
        const uint32_t i = min(i0 + u * 256u, last);
        c[u] = cell_of_atom[i]; rk[u] = rank_of_atom[i];
        x[u] = in.x[i]; y[u] = in.y[i]; z[u] = in.z[i];
        at[u] = in.attr[i]; ro[u] = in.res_ord[i]; cr[u] = in.chain_rank[i]; md[u] = in.model[i];
        rid[u] = any_h ? in.res_id[i] : 0u;
    }
    uint32_t d[kPlacePer], h0[kPlacePer], h1[kPlacePer];
    double mx[kPlacePer], my[kPlacePer], mz[kPlacePer];
#pragma unroll
    for (uint32_t u = 0; u < kPlacePer; u++) {  // the dependent loads, again all together
        const bool use = (i0 + u * 256u <= last) & (c[u] != ARP_NONE);
        if (!use) c[u] = ARP_NONE;
        d[u] = use ? cell_start[c[u]] : 0u;
        h0[u] = h1[u] = 0u;
        if (any_h) { h0[u] = in.res_h_ptr[rid[u]]; h1[u] = in.res_h_ptr[rid[u] + 1u]; }
        mx[u] = gmx; my[u] = gmy; mz[u] = gmz;
        if (morg) { const double *o = morg + 6u * min(md[u], kPackModels - 1u) + 3u; mx[u] = o[0]; my[u] = o[1]; mz[u] = o[2]; }  // packed batch: the member's own midpoint
    }
#pragma unroll
    for (uint32_t u = 0; u < kPlacePer; u++) {
        if (c[u] == ARP_NONE) continue;
        const uint32_t slot = d[u] + rk[u], i = i0 + u * 256u;
        const float fx = (float)(x[u] - mx[u]), fy = (float)(y[u] - my[u]), fz = (float)(z[u] - mz[u]);
        so.rec[slot] = make_float4(fx, fy, fz, (float)((double)fx * fx + (double)fy * fy + (double)fz * fz));
        // "the residue carries hydrogens" as a bit of the record: the hot kernel never touches the hydrogen tables, the deferred
        // pass resolves residue -> hydrogens itself (hbond.rs:38-42)
        const bool has_h = h0[u] < h1[u];
        const uint32_t attr = at[u] & ~kAttrResHasH;
        Fat f;
        f.x = x[u]; f.y = y[u]; f.z = z[u];
        f.pw = make_pair_word(attr, has_h); f.res_ord = ro[u];
        f.crm = cr[u]; f.orig = i; f.cell = c[u];
        f.attr = attr | (has_h ? kAttrResHasH : 0u);
        so.fat[slot] = f;
        if (RKEY) {
            bool bad;
            so.rkey[slot] = residue_word(ro[u], cr[u], &bad);
            // (an input of 70 000 chains would otherwise queue 70 000 atomics on one word: the flag is read first, so only the first few set it)
            if (bad && __hip_atomic_load(&gp->rk_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(&gp->rk_bad, 1u);
        }
    }
}

__global__ __launch_bounds__(256) void k_scatter(uint32_t n, const uint32_t *cell_of_atom, const uint32_t *rank_of_atom,
                                                 const uint32_t *cell_start, uint32_t *perm, uint32_t *slot_cell) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = cell_of_atom[i];
    if (c == ARP_NONE) return;
    uint32_t p = cell_start[c] + rank_of_atom[i];
    perm[p] = i;
    slot_cell[p] = c;
}

// Ordered mode: final slot = cell_start + rank of the atom index inside its cell, so the sorted order (and therefore the
// order of the emitted pairs) does not depend on the arrival order of the atomics in k_cellid.
__global__ __launch_bounds__(256) void k_gather(DevAtoms in, GridParams *gp, const uint32_t *cell_start, const uint32_t *perm,
                                                const uint32_t *slot_cell, Sorted so) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ncells = gp->ncells;
    uint32_t n_heavy = cell_start[ncells];
    if (p == 0) { gp->n_heavy = n_heavy; gp->n_tasks = (n_heavy + 63u) / 64u; }
    if (p >= n_heavy) return;
    uint32_t c = slot_cell[p], i = perm[p];
    uint32_t s = cell_start[c], e = cell_start[c + 1], rank = 0;
    for (uint32_t q = s; q < e; q += 4) {  // four independent loads per trip: the loop is latency-bound otherwise
        const uint32_t u0 = perm[q], u1 = (q + 1 < e) ? perm[q + 1] : 0xFFFFFFFFu, u2 = (q + 2 < e) ? perm[q + 2] : 0xFFFFFFFFu,
                       u3 = (q + 3 < e) ? perm[q + 3] : 0xFFFFFFFFu;
        rank += (u0 < i) + (u1 < i) + (u2 < i) + (u3 < i);
    }
    place_atom(in, gp, so, i, c, s + rank);
}
