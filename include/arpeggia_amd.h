/*
 * arpeggia_amd.h -- C ABI of the MI355X-native contact engine (libarpeggia_amd.so).
 *
 * Drop-in boundary for ONE path of y1zhou/arpeggia v0.8.0: `arpeggia::get_contacts`
 * (src/contacts/mod.rs:61, re-exported src/lib.rs:28) and, inside it, the hot loop
 * `Interactions::get_atomic_contacts` (src/contacts/complex.rs:189-299).  The reference has no FFI of its
 * own (pure Rust); these are the entry points a Rust `extern "C"` block / ctypes stub binds -- see
 * INTEGRATION.md for the exact reference-side stubs.  Plain pointers and sizes only, no torch/HIP types:
 * a HIP stream crosses as `void*`.
 *
 * Every function returns an arp_status and never unwinds across the boundary (the reference panics instead:
 * utils.rs:77,109; complex.rs:50; vdw.rs:58-69).  arp_last_error() gives the thread-local message, which
 * reproduces the reference's panic strings where its tests pin them (utils.rs:215,223).
 */
#ifndef ARPEGGIA_AMD_H
#define ARPEGGIA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARP_API_VERSION 2   /* v2 (round 4): arp_atoms.chain_rank and arp_atoms.model are 32-bit (v1: uint16_t, at most 65 535 chains / models) */

typedef int32_t arp_status;
enum {
    ARP_OK = 0,
    ARP_ERR_BAD_GROUPS = 1,   /* utils.rs:77  "Invalid chain groups format! Use '/' for all-to-all comparisons." */
    ARP_ERR_EMPTY_GROUPS = 2, /* utils.rs:109 "Empty chain groups!" */
    ARP_ERR_NO_RINGS = 3,     /* complex.rs:50 "Error building ring positions" (no HIS/PHE/TYR/TRP ring at all) */
    ARP_ERR_BAD_INPUT = 4,    /* malformed arrays / unsupported element / CYS without CB (vdw.rs:58 unwrap) ... */
    ARP_ERR_HIP = 5,          /* a HIP runtime call failed */
    ARP_ERR_OOM = 6,
    ARP_ERR_NO_DEVICE = 7,    /* no gfx950 device visible: the engine has NO CPU fallback */
    ARP_ERR_IO = 8,
    ARP_ERR_CAPACITY = 9      /* caller-provided pair buffer too small; required size reported */
};

/* ---- interaction vocabulary: bit k of arp_pair.kind <=> variant k of the reference enum (structs.rs:6-51) ---- */
enum {
    ARP_StericClash = 0, ARP_CovalentBond, ARP_Disulfide, ARP_VanDerWaalsContact, ARP_IonicBond, ARP_HydrogenBond,
    ARP_WeakHydrogenBond, ARP_PolarContact, ARP_WeakPolarContact, ARP_IonicRepulsion, ARP_SaltBridge,
    ARP_PiDisplacedStacking, ARP_PiTStacking, ARP_PiSandwichStacking, ARP_PiParallelInPlaneStacking,
    ARP_PiTiltedStacking, ARP_PiLStacking, ARP_CationPi, ARP_HydrophobicContact, ARP_N_INTERACTIONS
};

/* ---- per-atom attribute word (arp_atoms.attr) ---- */
#define ARP_ATTR_ELEM_MASK   0x0000000Fu /* element class: index into arp_params.cov_radius / vdw_radius            */
#define ARP_ATTR_DONOR       0x00000010u /* hbond.rs:160-178 is_hydrogen_donor (conformer name, atom name)          */
#define ARP_ATTR_ACCEPTOR    0x00000020u /* hbond.rs:137-157 is_hydrogen_acceptor                                    */
#define ARP_ATTR_WEAK_DONOR  0x00000040u /* hbond.rs:204-207 element C and name != "C"                               */
#define ARP_ATTR_POS         0x00000080u /* ionic.rs:84-91 (conformer name)                                          */
#define ARP_ATTR_NEG         0x00000100u /* ionic.rs:94-99                                                           */
#define ARP_ATTR_HYDROPHOBIC 0x00000200u /* hydrophobic.rs:27-45 (residue name)                                      */
#define ARP_ATTR_CYS_SG      0x00000400u /* vdw.rs:50-53: residue CYS and atom SG                                    */
#define ARP_ATTR_H           0x00000800u /* element H: never a candidate (complex.rs:83-87,201), only an H-bond probe */
#define ARP_ATTR_LIGAND      0x00001000u /* chain in the ligand set   (utils.rs:71-115)                              */
#define ARP_ATTR_RECEPTOR    0x00002000u /* chain in the receptor set                                                */
#define ARP_ATTR_POS_RESN    0x00004000u /* ionic.rs:84-91 keyed by RESIDUE name (cation-pi, aromatic.rs:18)         */

#define ARP_NONE 0xFFFFFFFFu

enum { ARP_MEM_HOST = 0, ARP_MEM_DEVICE = 1 };

/* SoA view of one structure, borrowed for the duration of a call (the reference borrows &PDB: mod.rs:61).
 * All arrays have n entries unless noted.  `location` says where EVERY pointer lives. */
typedef struct arp_atoms {
    uint64_t n;                  /* atoms, hydrogens included                                                 */
    const double *x, *y, *z;     /* f64 coordinates (pdbtbx Atom::pos)                                        */
    const uint32_t *attr;        /* ARP_ATTR_* bits                                                           */
    const uint32_t *res_ord;     /* positional index of the residue in its chain (complex.rs:411-440)         */
    const uint32_t *chain_rank;  /* rank of the chain id under byte-wise string order (complex.rs:129); the reference keys on the id
                                  * STRING (complex.rs:19-21): any number of chains                                     */
    const uint32_t *model;       /* model ordinal 0, 1, 2, ... (complex.rs:96-98 same-model test); dense: every model owns a
                                  * slab of the cell grid, the largest ordinal is bounded by the grid (~4 per atom)             */
    /* tables for the rare data-dependent rules; may be NULL when n_res == 0 (then no H probes, no disulfides) */
    const uint32_t *res_id;      /* per atom: global residue ordinal                                          */
    uint64_t n_res;
    const uint32_t *res_h_ptr;   /* n_res+1: CSR residue -> hydrogen atoms (hbond.rs:38-42 scans the residue) */
    const uint32_t *res_h_idx;   /* atom indices of the hydrogens                                             */
    const uint32_t *res_cb;      /* n_res: first CB of the residue or ARP_NONE (vdw.rs:55-58)                 */
    const uint32_t *res_sg;      /* n_res: first SG of the residue or ARP_NONE (vdw.rs:59-63)                 */
    int32_t location;            /* ARP_MEM_HOST | ARP_MEM_DEVICE                                             */
    int32_t reserved;
} arp_atoms;

/* arp_params.flags.  By default pairs are emitted in one pass in an unspecified order (like the reference, whose order is
 * that of an R*-tree walk + rayon, complex.rs:194-298).  DETERMINISTIC selects the two-pass count/scan/fill emitter
 * whose output order is a function of the input only (about 2x slower: 0.54 against 0.26 ms on 10^6 atoms).
 * CONTACTS_ONLY drops the candidates no rule matched (kind == 0) on the device: what is left is exactly the set of
 * pairs get_atomic_contacts turns into ResultEntry rows (complex.rs:208-297), typically 5-10% of the candidates, so the
 * copy to the host and the table assembly shrink by that factor.  arp_get_contacts uses it. */
#define ARP_FLAG_DETERMINISTIC 0x1u
#define ARP_FLAG_CONTACTS_ONLY 0x2u
/* NO_SPECULATION (arp_contacts_atomic_enqueue): the single-pass emitter normally skips the launch of the probe pass (hydrogen-bond angles,
 * disulfide dihedrals) when the previous call on the same arrays needed none, checks the guess on the device, and lets
 * arp_contacts_atomic_result repeat the whole call when the guess was wrong -- until then the records a probe decides sit in `out` with
 * kind 0 (inputs below ~20 000 atoms run their probes inside the emitter: nothing is speculated there; between ~20 000 and ~131 000 atoms the same
 * guess also selects a launch sequence without the hole fix-up -- if it was wrong nothing of that call's output is valid until the repeat has run).  With this flag the probe pass is always launched behind the emitter, on the same stream: work the caller orders on that stream
 * after the enqueue sees final records.  (One repeat remains possible, for either setting: a deferred-probe list that overflows -- an input
 * with more than ~16 probe candidates per atom -- is grown by arp_contacts_atomic_result and the call run again; it then returns only after
 * the repeat, and what ran on the stream in between has seen an incomplete list.  ARP_FLAG_DETERMINISTIC never speculates.) */
#define ARP_FLAG_NO_SPECULATION 0x4u
/* RESIDUE_RUNS / NO_RESIDUE_RUNS: a hint about the input, never a change of the result.  The reference never pairs two atoms of one residue or
 * of sequence neighbours in one chain (complex.rs:108-113); in an input whose residues are runs of atoms (every protein) a third of an atom's
 * geometric neighbours are such atoms, and the single-pass emitter has kernels that drop them before the exact phase.  They cost a little on an
 * input of one-atom residues (a synthetic cloud), so the engine picks them from a sample of the PREVIOUS call's atoms on the same context
 * (how many of the first 255 atoms continue their predecessor's residue).  RESIDUE_RUNS asks for them outright (a first call, mixed workloads),
 * NO_RESIDUE_RUNS rules them out.  Both kernels emit the same list. */
#define ARP_FLAG_RESIDUE_RUNS 0x8u
#define ARP_FLAG_NO_RESIDUE_RUNS 0x10u

typedef struct arp_params {
    double vdw_comp;             /* mod.rs:61 vdw_comp    (default 0.1) */
    double dist_cutoff;          /* mod.rs:61 dist_cutoff (default 6.5) */
    double cov_radius[16];       /* by element class: pdbtbx covalent_single  (vdw.rs:24-28) */
    double vdw_radius[16];       /* by element class: pdbtbx van_der_waals                   */
    double h_vdw_radius;         /* Element::H van_der_waals (hbond.rs:52)                   */
    uint32_t flags;              /* ARP_FLAG_* */
    uint32_t reserved;
} arp_params;

/* One classified candidate pair = one element of `ligand_neighbors` (complex.rs:194-213) after the per-pair
 * rules (complex.rs:215-298).  i = ligand atom x, j = receptor atom y (indices into arp_atoms).  Pairs with
 * kind == 0 are candidates that produced no row. */
typedef struct arp_pair {
    uint32_t i, j;
    float dist;                  /* (f32) Atom::distance, as the table stores it (mod.rs:148) */
    uint32_t kind;               /* bit set over ARP_* interaction codes */
} arp_pair;

typedef struct arp_pairs {
    uint64_t n;
    arp_pair *data;              /* owned by the library until arp_pairs_free */
    int32_t location;            /* where data lives */
    int32_t reserved;
} arp_pairs;

typedef struct arp_context arp_context;     /* one per (device, stream); owns the reusable workspace */
typedef struct arp_structure arp_structure; /* parsed + filtered model (utils.rs:51-63 load_model)   */
typedef struct arp_table arp_table;         /* the 20-column contact table (mod.rs:140-214)          */

/* ---- library / device ---- */
/* Diagnostics -- not part of the reference's surface.  The library reads ONE environment variable, ARPEGGIA_AMD_HOST_POOL_MB (idle pinned host
 * memory kept for the next batch / table, default 4096); every other switch is set here, process-wide:
 *   "timing"        1: stage laps of the table, batch and ingest paths on stderr
 *   "emit_kernel"   1: the single-pass emitter runs its alternative kernel (both operands gathered; the route of inputs beyond 2^24 slots), so that
 *                      the parity suite can check it on ordinary inputs; 0 (default): chosen by input size
 *   "defer_entries" N > 0: entries of the deferred-probe list of workspaces allocated from now on (a tiny list makes the grow-and-repeat path run)
 *   "strip_rows"    N = a power of two: the cell rows of single-model inputs are ordered in y strips of N rows (chosen by input size when 0, the default:
 *                      strips only from ~2.5 x 10^6 atoms of a compact structure on), for parameter blocks built from now on -- lets the parity suite run the strip order on small inputs
 *   "table_host"    1: only in the test library built with -DARP_WITH_HOST_TABLE (tests/hosttable): arp_get_contacts assembles the table on the host
 * Unknown keys return ARP_ERR_BAD_INPUT. */
arp_status arp_debug_set(const char *key, int64_t value);
int32_t arp_api_version(void);
/* A binder compiled against this header calls arp_check_api_version(ARP_API_VERSION) once: ARP_OK when the library lays out arp_atoms /
 * arp_params / arp_pair as that version of the header does, ARP_ERR_BAD_INPUT (+ arp_last_error) otherwise -- a v1 caller (16-bit chain
 * ranks and models) would otherwise hand over arrays the v2 kernels read 4 bytes per atom. */
arp_status arp_check_api_version(int32_t header_version);
const char *arp_strerror(arp_status s);
const char *arp_last_error(void);
int32_t arp_device_count(void);             /* gfx950 devices visible; 0 => every compute call fails with ARP_ERR_NO_DEVICE */
const char *arp_interaction_name(int32_t code); /* structs.rs:151-157 Display == variant name */
void arp_default_params(arp_params *p);     /* 0.1 / 6.5 and the radii of the built-in element classes */
int32_t arp_element_class(const char *symbol); /* class index used by arp_default_params, -1 if unsupported */

/* ---- context ---- */
arp_status arp_context_create(int32_t device, arp_context **out);
void arp_context_destroy(arp_context *ctx);
/* Launch on a caller-owned HIP stream (hipStream_t passed as void*; NULL = the legacy default stream).  A new context
 * starts on a private non-blocking stream of its own. */
arp_status arp_context_set_stream(arp_context *ctx, void *hip_stream);
arp_status arp_context_synchronize(arp_context *ctx);

/* ---- the hot path: replaces Interactions::get_atomic_contacts (complex.rs:189-299) ---- */
/* Synchronous.  Inputs may be host or device arrays; output is library-allocated where `out_location` says. */
arp_status arp_contacts_atomic(arp_context *ctx, const arp_atoms *atoms, const arp_params *params,
                               int32_t out_location, arp_pairs *out);
void arp_pairs_free(arp_pairs *pairs);

/* Asynchronous, allocation-free form for resident data (inputs MUST be ARP_MEM_DEVICE): enqueues the whole
 * pipeline on the context's stream and returns.  `out` is a device buffer of `capacity` pairs.
 * arp_contacts_atomic_result() synchronises the stream and returns the pair count (ARP_ERR_CAPACITY + the required
 * count when the buffer was too small; nothing is written past `capacity`).
 * CONTRACT: the contents of `out` (and the count) are DEFINED ONLY AFTER arp_contacts_atomic_result HAS RETURNED ARP_OK.  Between the
 * two calls the buffer is speculative: the result call may run the enqueued work a second time (see ARP_FLAG_NO_SPECULATION), so
 * stream work ordered between enqueue and result must not consume `out` unless that flag is set -- and `atoms`, `params`' arrays and
 * `out` must stay alive and unchanged until the result call returns. */
arp_status arp_contacts_atomic_enqueue(arp_context *ctx, const arp_atoms *atoms, const arp_params *params,
                                       arp_pair *out, uint64_t capacity);
arp_status arp_contacts_atomic_result(arp_context *ctx, uint64_t *n_pairs);

/* Batch of independent structures sharded over devices (no collective).  ctxs[d] is a context on
 * device d; structure k goes to a device by longest-processing-time-first on its atom count; one host thread
 * per device; with ARP_FLAG_CONTACTS_ONLY small structures of a device's share are packed into shared launches.
 * outs[k] is filled like arp_contacts_atomic (host memory).  The lists of structures that shared a pack are views into ONE pinned block
 * (one PCIe copy per pack, no copy per member): release every outs[k] with arp_pairs_free as usual -- the block is reference-counted and
 * goes back to a pool when its last list is freed; do not free() the pointers yourself. */
arp_status arp_contacts_atomic_batch(arp_context *const *ctxs, int32_t n_ctx, const arp_atoms *const *atoms,
                                     int32_t n_structures, const arp_params *params, arp_pairs *outs);
/* Pinned host blocks whose last pair list or table has been freed are kept for the next batch / table (pinning memory costs far more than
 * the copy it saves; at most 4 GiB stay pooled -- ARPEGGIA_AMD_HOST_POOL_MB in the environment sets another limit, 0 keeps nothing -- and a
 * request only reuses a pooled block of at most twice its size).  This returns the idle ones to the system; blocks still referenced are untouched.
 * Returns the number of bytes released. */
uint64_t arp_release_host_pool(void);

/* ---- SAP neighbour sum: the radius sum of src/sap.rs:155-204 on the same cell list ----
 * out[i] = sum over the atoms j with sidechain[j] != 0 and |r_j - r_i|^2 <= f64(sap_radius * sap_radius) (inclusive, i itself included) of
 * weight[j], accumulated in f32, for every atom i with sidechain[i] != 0; 0 for the others.  Host arrays.  The per-atom SASA behind the
 * weights (src/sasa.rs, rust-sasa) is the caller's: arp_sap_weight gives hydrophobicity(resn) * clamp(sasa / max_sc_asa(resn), 0, 1) as the
 * reference forms it (sap.rs:41-101,198-209). */
float arp_sap_weight(const char *resn, float sasa);
arp_status arp_sap_neighbor_sum(arp_context *ctx, uint64_t n, const double *x, const double *y, const double *z,
                                const uint8_t *sidechain, const float *weight, float sap_radius, float *out);

/* Per-kernel device timing of the most recent call (HIP events on the context's stream).  Enable, run, then read.
 * names[k] points to a static string.  Returns the number of kernels recorded (<= cap). */
arp_status arp_profile_enable(arp_context *ctx, int32_t on);
int32_t arp_profile_read(arp_context *ctx, const char **names, float *milliseconds, int32_t cap);

/* ---- structure ingest: replaces utils.rs:51-63 load_model (+ python.rs:45-47) ---- */
arp_status arp_structure_load(const char *path, int32_t ignore_zero_occupancy, arp_structure **out);
/* Build from flat per-atom records (fixed-width, NUL-padded strings).  hierarchy == 0: derive the
 * Model>Chain>Residue>Conformer hierarchy the way pdbtbx does and apply the load_model residue filter.
 * hierarchy == 1: take res_ord / res_id as given (synthetic SoA inputs). */
typedef struct arp_records {
    uint64_t n;
    const double *x, *y, *z, *occupancy;
    const int32_t *serial, *resi, *model_serial;
    const char *name;      /* n x 8  atom name          */
    const char *resn;      /* n x 8  conformer name     */
    const char *chain;     /* n x 8  chain id           */
    const char *altloc;    /* n x 4                     */
    const char *icode;     /* n x 4  insertion code     */
    const char *element;   /* n x 4                     */
    const uint32_t *res_ord; /* hierarchy == 1 only */
    const uint32_t *res_id;  /* hierarchy == 1 only */
} arp_records;
arp_status arp_structure_from_records(const arp_records *rec, int32_t hierarchy, arp_structure **out);
void arp_structure_free(arp_structure *s);
uint64_t arp_structure_n_atoms(const arp_structure *s);
/* Host SoA view for a given chain grouping (utils.rs:71-115 parse_groups sets the LIGAND/RECEPTOR bits).
 * The view stays valid until the next arp_structure_atoms call on `s` or arp_structure_free. */
arp_status arp_structure_atoms(arp_structure *s, const char *groups, arp_atoms *view);
/* Per-atom identity columns, n x width fixed-width strings / n ints (valid while `s` lives). */
const char *arp_structure_strings(const arp_structure *s, const char *column, int32_t *width);
const int32_t *arp_structure_ints(const arp_structure *s, const char *column);

/* Host worker threads of the table path (entity bookkeeping, copies out of pinned memory, column / Arrow materialisation -- the plane
 * fits, ring rows, row assembly and sort run on the device): the reference's global rayon pool (utils.rs:8-30; `num_threads` of
 * python.rs:31).  1 = serial (default, as in the reference), 0 = all hardware threads.  The table is identical for every count. */
void arp_set_num_threads(int32_t n);
int32_t arp_get_num_threads(void);

/* ---- the table: replaces arpeggia::get_contacts (mod.rs:61-137) ----
 * The structure is kept resident on the context's device (uploaded once, with its fitted planes and entity ranks); calls on ONE
 * arp_structure are serialised inside the library, different structures (on different contexts) run in parallel. */
arp_status arp_get_contacts(arp_context *ctx, arp_structure *s, const char *groups, double vdw_comp,
                            double dist_cutoff, arp_table **out);
/* The same with the host worker count of THIS call given explicitly (the reference sizes a scoped rayon pool per call,
 * utils.rs:8-30): num_threads > 0 that many, 0 all hardware threads, < 0 the process-wide default of arp_set_num_threads.
 * Concurrent calls with different counts do not interfere. */
arp_status arp_get_contacts_mt(arp_context *ctx, arp_structure *s, const char *groups, double vdw_comp,
                               double dist_cutoff, int32_t num_threads, arp_table **out);
/* The ring / side-chain planes as the device fits them (residues.rs:270-298), per residue of the filtered model in
 * hierarchy order: planes = n_residues x 12 doubles {ring centre, ring normal, sc centre, sc normal}; valid[r] bit 1 = the residue has a
 * ring plane, bit 2 = a side-chain plane. */
uint64_t arp_structure_n_residues(const arp_structure *s);
arp_status arp_structure_fit_planes(arp_context *ctx, arp_structure *s, double *planes, uint8_t *valid);
void arp_table_free(arp_table *t);
uint64_t arp_table_rows(const arp_table *t);
/* Column by reference name (mod.rs:140-181,209-211): "model" u32; "interaction" i32 code; "distance" f32;
 * "from_resi"/"from_atomi"/"to_resi"/"to_atomi" i32; "sc_centroid_dist"/"sc_dihedral"/"sc_centroid_angle" f32
 * (+ "sc_valid" u8: 0 => null); string columns are rows x width fixed-width NUL-padded chars.  Also
 * "from_atom"/"to_atom" i32 atom indices (-1 for a "Ring" entity). */
const void *arp_table_column(const arp_table *t, const char *name, int32_t *width);

/* The same 20 columns through the Arrow C Data Interface (a struct array = one record batch; utf8 strings, nullable
 * f32 sc_* columns): what pyo3-polars hands to Python in the reference (python.rs:55, mod.rs:140-214), importable with
 * zero per-row work by pyarrow / polars / arrow-rs (`FFI_ArrowArray`).  The exported arrays own copies of the columns:
 * the table may be freed before they are released.  Struct layout per the Arrow specification (ABI-stable). */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char *format, *name, *metadata;
    int64_t flags, n_children;
    struct ArrowSchema **children, *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length, null_count, offset, n_buffers, n_children;
    const void **buffers;
    struct ArrowArray **children, *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif
arp_status arp_table_export_arrow(const arp_table *t, struct ArrowArray *out_array, struct ArrowSchema *out_schema);

#ifdef __cplusplus
}
#endif
#endif
