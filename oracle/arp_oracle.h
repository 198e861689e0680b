/*
 * arp_oracle.h -- CPU ORACLE (test infrastructure only) for the arpeggia `contacts` hot path.
 *
 * This is a from-scratch plain-C restatement of the reference algorithm
 * (y1zhou/arpeggia v0.8.0, src/contacts/ and src/utils.rs).  It exists ONLY to
 * check the HIP product path: it may be imported/linked by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the
 * product (arpeggia_amd/).  It is deliberately simple: string rules are applied
 * literally per pair, exactly as the reference does, and everything is f64.
 *
 * Parity pinning status: see the header comment of arp_oracle.c.
 */
#ifndef ARP_ORACLE_H
#define ARP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One atom record.  Layout is mirrored by a numpy structured dtype in
 * tests/oracle_binding.py (checked against orc_sizeof_atom()). */
typedef struct OrcAtom {
    double x, y, z, occ;
    int32_t serial;        /* file atom serial ("atomi" in the table)           */
    int32_t resi;          /* residue sequence number                            */
    int32_t model_serial;  /* MODEL record number, 0 when the file has none      */
    int32_t hetero;        /* 1 for HETATM                                       */
    char name[8];          /* atom name, trimmed, upper-cased                    */
    char resn[8];          /* conformer name (3-letter residue name of the line) */
    char chain[8];         /* chain id, trimmed                                  */
    char altloc[4];        /* "" when blank                                      */
    char icode[4];         /* insertion code, "" when blank                      */
    char elem[4];          /* element symbol, upper-cased                        */
    /* hierarchy, assigned by orc_from_atoms(flat=0) / orc_load_model, or given by the caller (flat=1) */
    int32_t model_idx;     /* ordinal of the model                               */
    int32_t chain_idx;     /* global ordinal of the (model, chain)               */
    int32_t res_idx;       /* global ordinal of the residue (model, chain, resi, icode) */
    int32_t res_ord;       /* positional index of the residue within its chain (complex.rs:411-440) */
    char res_resn[8];      /* residue().name(): common name of all conformers    */
} OrcAtom;

/* One classified candidate pair: x = ligand atom index, y = receptor atom index. */
typedef struct OrcPair {
    int32_t i, j;
    double dist;
    uint32_t kind;         /* one bit per Interaction variant, ORC_* below */
    uint32_t pad;
} OrcPair;

/* Interaction codes = position in the reference enum (structs.rs:6-51). */
enum {
    ORC_StericClash = 0, ORC_CovalentBond, ORC_Disulfide, ORC_VanDerWaalsContact,
    ORC_IonicBond, ORC_HydrogenBond, ORC_WeakHydrogenBond, ORC_PolarContact,
    ORC_WeakPolarContact, ORC_IonicRepulsion, ORC_SaltBridge,
    ORC_PiDisplacedStacking, ORC_PiTStacking, ORC_PiSandwichStacking,
    ORC_PiParallelInPlaneStacking, ORC_PiTiltedStacking, ORC_PiLStacking, ORC_CationPi,
    ORC_HydrophobicContact, ORC_N_INTERACTIONS
};

typedef struct OrcEntity {
    char chain[8], resn[8], insertion[4], altloc[4], atomn[8];
    int32_t resi, atomi;
} OrcEntity;

/* One row of the 20-column contact table (mod.rs:140-214). */
typedef struct OrcRow {
    uint32_t model;
    int32_t interaction;
    double distance;           /* f64; the table column is (float)distance */
    OrcEntity from, to;
    int32_t has_sc;            /* 0 => the three sc_* columns are null */
    double sc_centroid_dist, sc_dihedral, sc_centroid_angle;
    int32_t from_atom, to_atom; /* atom indices (-1 for a "Ring" entity) */
} OrcRow;

typedef struct OrcPlane {
    double c[3], n[3];
    int32_t model_serial, resi, res_idx, res_ord;
    char chain[8], resn[8], insertion[4], altloc[4];
} OrcPlane;

typedef struct OrcStructure OrcStructure;

enum { ORC_OK = 0, ORC_ERR_IO = 1, ORC_ERR_BAD_GROUPS = 2, ORC_ERR_EMPTY_GROUPS = 3,
       ORC_ERR_NO_RINGS = 4, ORC_ERR_BAD_INPUT = 5, ORC_ERR_OOM = 6 };

int orc_sizeof_atom(void);
int orc_sizeof_pair(void);
int orc_sizeof_row(void);
int orc_sizeof_plane(void);
const char *orc_interaction_name(int code);
const char *orc_last_error(void);

/* utils.rs:51-63 load_model (+ python.rs:45-47 zero-occupancy strip). NULL on error. */
OrcStructure *orc_load_model(const char *path, int ignore_zero_occupancy);
/* Build from records.  flat=0: construct the pdbtbx-like hierarchy and apply the load_model residue
 * filter.  flat=1: trust model_idx/chain_idx/res_idx/res_ord/res_resn as given (synthetic SoA inputs). */
OrcStructure *orc_from_atoms(const OrcAtom *atoms, int32_t n, int flat);
void orc_free_structure(OrcStructure *s);
void orc_free(void *p);

int32_t orc_n_atoms(const OrcStructure *s);
const OrcAtom *orc_atoms(const OrcStructure *s);

/* utils.rs:71-115.  Writes NUL-separated sorted chain ids; returns ORC_* status. */
int orc_parse_groups(const char *const *all_chains, int n_chains, const char *groups,
                     char *lig_out, int lig_cap, int *n_lig, char *rec_out, int rec_cap, int *n_rec);

/* complex.rs:189-299.  mode 0 = uniform grid search, 1 = brute force O(N^2).
 * Returns every candidate pair (kind may be 0), unordered. */
int orc_atomic_contacts(const OrcStructure *s, const char *groups, double vdw_comp, double dist_cutoff,
                        int mode, OrcPair **pairs, int64_t *n_pairs);

/* complex.rs:442-514: ring planes / side-chain planes.  which: 0 = rings, 1 = sc planes. */
int orc_planes(const OrcStructure *s, int which, OrcPlane **planes, int32_t *n_planes);

/* mod.rs:61-137: the full sorted 20-column table. */
int orc_get_contacts(const OrcStructure *s, const char *groups, double vdw_comp, double dist_cutoff,
                     OrcRow **rows, int64_t *n_rows);

/* per-atom class predicates exposed for host-logic tests (string rules of hbond.rs/ionic.rs/hydrophobic.rs) */
uint32_t orc_atom_classes(const OrcAtom *a);
enum { ORC_CLS_DONOR = 1, ORC_CLS_ACCEPTOR = 2, ORC_CLS_WEAK_DONOR = 4, ORC_CLS_POS = 8, ORC_CLS_NEG = 16,
       ORC_CLS_HYDROPHOBIC = 32, ORC_CLS_CYS_SG = 64, ORC_CLS_H = 128, ORC_CLS_POS_RESN = 256 };

/* pdbtbx Atom::angle / Atom::dihedral restatements (degrees) */
double orc_angle(const double a[3], const double b[3], const double c[3]);
double orc_dihedral(const double a[3], const double b[3], const double c[3], const double d[3]);
int orc_radii(const char *elem, double *cov_single, double *vdw);
/* SAP neighbour sum (src/sap.rs:155-204), SASA supplied by the caller: see arp_oracle.c */
float orc_sap_weight(const char *resn, float sasa);
void orc_sap_neighbor_sum(int64_t n, const double *x, const double *y, const double *z, const uint8_t *sidechain, const float *weight, float sap_radius,
                          float *out);
void orc_plane_metrics(const double c1[3], const double n1[3], const double c2[3], const double n2[3], double out[3]);

#ifdef __cplusplus
}
#endif
#endif
