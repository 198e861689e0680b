/*
 * arp_oracle.c -- CPU ORACLE for the arpeggia `contacts` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product
 * (arpeggia_amd/, libarpeggia_amd.so) never links, imports or calls it.
 *
 * What it restates (y1zhou/arpeggia v0.8.0; all paths relative to the reference checkout):
 *   src/utils.rs:51-63      load_model      (Loose PDB read, keep only 20 aa + HOH residues)
 *   src/utils.rs:71-115     parse_groups
 *   src/contacts/complex.rs:76-131   should_compare_entities / should_compare_residues
 *   src/contacts/complex.rs:189-299  get_atomic_contacts (candidate generation + per-pair rules)
 *   src/contacts/complex.rs:301-405  get_ring_atom_contacts / get_ring_ring_contacts
 *   src/contacts/complex.rs:137-174,411-514  sc stats, residue index, ring / sc-plane tables
 *   src/contacts/vdw.rs, hbond.rs, ionic.rs, hydrophobic.rs, aromatic.rs  (rule functions)
 *   src/contacts/residues.rs:24-75,163-298  Plane maths, ring / sc-plane atom tables, plane fit
 *   src/contacts/mod.rs:61-214  table assembly, sc-stat join, 10-key sort
 *
 * Third-party arithmetic that is NOT in /root/reference (no Cargo.lock, nothing vendored) and is
 * restated here from the crates' published behaviour:
 *   pdbtbx 0.12.0  PDB reader + hierarchy (Model>Chain>Residue>Conformer>Atom), Atom::distance/angle/
 *                  dihedral, Element::atomic_radius (vdW = Alvarez 2013, covalent = Pyykko-Atsumi 2009)
 *   rstar 0.12.2   locate_within_distance: inclusive d^2 <= r^2
 *   nalgebra 0.33  svd(): normal = left singular vector of the smallest singular value
 *   polars 0.52    concat / left join / ascending multi-key sort
 *
 * PARITY PINNING: the reference cannot be built or imported in this environment (Rust; no cargo/rustc;
 * the Python package is a compiled extension).  This oracle is pinned by the reference's OWN test
 * facts only (tests/test_oracle_golden.py):
 *   - contacts(1ubq, "/", 0.1, 6.5) has exactly 532 rows x 20 columns (python/tests/test_arpeggia.py:35,67)
 *   - 1ubq PHE4 ring centre / normal                              (src/contacts/residues.rs:355-372)
 *   - 6bft TYR A102 ring .. ARG G82 NE = CationPi; TRP A108 .. LYS G84 NZ = none (aromatic.rs:72-128)
 *   - Plane identities (residues.rs:306-332); parse_groups cases + panic strings (utils.rs:174-228)
 *   - 1ubq: 602 protein atoms + 58 waters in chain A; zero-occupancy strip is a no-op
 * Everything those facts do not reach (hydrogen-dependent branches, covalent radii, disulfide window,
 * pi-pi classes, row contents/order, multi-model / altloc behaviour) is "parity unpinned".
 */
#define _GNU_SOURCE
#include "arp_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ errors */
static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
static void set_err(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

int orc_sizeof_atom(void) { return (int)sizeof(OrcAtom); }
int orc_sizeof_pair(void) { return (int)sizeof(OrcPair); }
int orc_sizeof_row(void) { return (int)sizeof(OrcRow); }
int orc_sizeof_plane(void) { return (int)sizeof(OrcPlane); }
void orc_free(void *p) { free(p); }

static const char *k_names[ORC_N_INTERACTIONS] = {
    "StericClash", "CovalentBond", "Disulfide", "VanDerWaalsContact", "IonicBond", "HydrogenBond",
    "WeakHydrogenBond", "PolarContact", "WeakPolarContact", "IonicRepulsion", "SaltBridge",
    "PiDisplacedStacking", "PiTStacking", "PiSandwichStacking", "PiParallelInPlaneStacking",
    "PiTiltedStacking", "PiLStacking", "CationPi", "HydrophobicContact"};
/* structs.rs:151-157: Display == Debug variant name */
const char *orc_interaction_name(int code) {
    return (code >= 0 && code < ORC_N_INTERACTIONS) ? k_names[code] : "?";
}

/* ------------------------------------------------------------------ structure */
typedef struct {
    int32_t model_idx, model_serial;
    char id[8];
} OrcChain;

struct OrcStructure {
    OrcAtom *atoms;
    int32_t n;
    int32_t *conf_ord;     /* per atom: ordinal of its conformer inside its residue (hierarchy order) */
    OrcChain *chains;      /* every chain of every model, also chains emptied by the residue filter */
    int32_t n_chains;
    int32_t n_res;
    /* residue -> atoms CSR in hierarchy order (conformer ordinal, then file order) */
    int32_t *res_ptr, *res_atoms;
};

int32_t orc_n_atoms(const OrcStructure *s) { return s->n; }
const OrcAtom *orc_atoms(const OrcStructure *s) { return s->atoms; }

void orc_free_structure(OrcStructure *s) {
    if (!s) return;
    free(s->atoms); free(s->conf_ord); free(s->chains); free(s->res_ptr); free(s->res_atoms);
    free(s);
}

static int streq(const char *a, const char *b) { return strcmp(a, b) == 0; }

/* ------------------------------------------------------------------ radii (pdbtbx Element::atomic_radius) */
typedef struct { const char *sym; double cov, vdw; } Radii;
/* covalent_single: Pyykko & Atsumi 2009; van_der_waals: Alvarez 2013.  C/N/O/S vdW values are pinned by
 * the reference's 532-row test; the rest is unpinned. */
static const Radii k_radii[] = {
    {"H", 0.32, 1.20}, {"HE", 0.46, 1.43}, {"LI", 1.33, 2.12}, {"BE", 1.02, 1.98}, {"B", 0.85, 1.91},
    {"C", 0.75, 1.77}, {"N", 0.71, 1.66}, {"O", 0.63, 1.50}, {"F", 0.64, 1.46}, {"NE", 0.67, 1.58},
    {"NA", 1.55, 2.50}, {"MG", 1.39, 2.51}, {"AL", 1.26, 2.25}, {"SI", 1.16, 2.19}, {"P", 1.11, 1.90},
    {"S", 1.03, 1.89}, {"CL", 0.99, 1.82}, {"AR", 0.96, 1.83}, {"K", 1.96, 2.73}, {"CA", 1.71, 2.62},
    {"MN", 1.19, 2.45}, {"FE", 1.16, 2.44}, {"CO", 1.11, 2.40}, {"NI", 1.10, 2.40}, {"CU", 1.12, 2.38},
    {"ZN", 1.18, 2.39}, {"SE", 1.16, 1.82}, {"BR", 1.14, 1.86}, {"I", 1.33, 2.04}};

int orc_radii(const char *elem, double *cov, double *vdw) {
    for (size_t k = 0; k < sizeof k_radii / sizeof k_radii[0]; k++)
        if (streq(elem, k_radii[k].sym)) { *cov = k_radii[k].cov; *vdw = k_radii[k].vdw; return 1; }
    return 0;
}

/* ------------------------------------------------------------------ string rules */
static int in_list(const char *s, const char *const *list) {
    for (; *list; list++) if (streq(s, *list)) return 1;
    return 0;
}
#define LIST(...) ((const char *const[]){__VA_ARGS__, NULL})

/* residues.rs:131-161: the 20 standard amino acids + HOH survive load_model */
static int is_known_residue(const char *resn_upper) {
    return in_list(resn_upper, LIST("ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU",
                                    "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL", "HOH"));
}
/* hbond.rs:137-157 */
static int is_hydrogen_acceptor(const char *res, const char *atom) {
    if ((streq(atom, "O") || streq(atom, "OXT")) && !streq(res, "HOH")) return 1;
    if (streq(res, "ASN")) return streq(atom, "OD1");
    if (streq(res, "ASP")) return streq(atom, "OD1") || streq(atom, "OD2");
    if (streq(res, "GLN")) return streq(atom, "OE1");
    if (streq(res, "GLU")) return streq(atom, "OE1") || streq(atom, "OE2");
    if (streq(res, "HIS")) return streq(atom, "ND1") || streq(atom, "NE2");
    if (streq(res, "SER")) return streq(atom, "OG");
    if (streq(res, "THR")) return streq(atom, "OG1");
    if (streq(res, "TYR")) return streq(atom, "OH");
    if (streq(res, "MET")) return streq(atom, "SD");
    if (streq(res, "CYS")) return streq(atom, "SG");
    return 0;
}
/* hbond.rs:160-178 */
static int is_hydrogen_donor(const char *res, const char *atom) {
    if (streq(atom, "N")) return 1;
    if (streq(res, "ARG")) return in_list(atom, LIST("NE", "NH1", "NH2"));
    if (streq(res, "ASN")) return streq(atom, "ND2");
    if (streq(res, "GLN")) return streq(atom, "NE2");
    if (streq(res, "HIS")) return streq(atom, "ND1") || streq(atom, "NE2");
    if (streq(res, "LYS")) return streq(atom, "NZ");
    if (streq(res, "SER")) return streq(atom, "OG");
    if (streq(res, "THR")) return streq(atom, "OG1");
    if (streq(res, "TRP")) return streq(atom, "NE1");
    if (streq(res, "TYR")) return streq(atom, "OH");
    if (streq(res, "CYS")) return streq(atom, "SG");
    return 0;
}
/* hbond.rs:204-207 */
static int is_weak_hydrogen_donor(const OrcAtom *a) { return streq(a->elem, "C") && !streq(a->name, "C"); }
/* ionic.rs:84-91 */
static int is_pos_ionizable(const char *res, const char *atom) {
    if (streq(res, "ARG")) return in_list(atom, LIST("NE", "CZ", "NH1", "NH2"));
    if (streq(res, "HIS")) return in_list(atom, LIST("CG", "ND1", "CE1", "NE2", "CD2"));
    if (streq(res, "LYS")) return streq(atom, "NZ");
    return 0;
}
/* ionic.rs:94-99 */
static int is_neg_ionizable(const char *res, const char *atom) {
    if (streq(res, "ASP")) return streq(atom, "OD1") || streq(atom, "OD2");
    if (streq(res, "GLU")) return streq(atom, "OE1") || streq(atom, "OE2");
    return 0;
}
/* hydrophobic.rs:27-45 */
static int is_hydrophobic(const char *res, const char *atom) {
    if (streq(atom, "CB") && !streq(res, "SER")) return 1;
    if (in_list(res, LIST("ARG", "GLN", "GLU", "PRO"))) return streq(atom, "CG");
    if (streq(res, "ILE")) return in_list(atom, LIST("CG1", "CD1", "CG2"));
    if (streq(res, "LEU")) return in_list(atom, LIST("CG", "CD1", "CD2"));
    if (streq(res, "LYS")) return in_list(atom, LIST("CG", "CD"));
    if (streq(res, "MET")) return in_list(atom, LIST("CG", "CE", "SD"));
    if (streq(res, "PHE")) return in_list(atom, LIST("CG", "CD1", "CD2", "CE1", "CE2", "CZ"));
    if (streq(res, "THR")) return streq(atom, "CG2");
    if (streq(res, "TRP")) return in_list(atom, LIST("CG", "CD2", "CE3", "CZ3", "CH2", "CZ2"));
    if (streq(res, "TYR")) return in_list(atom, LIST("CG", "CD1", "CD2", "CE1", "CE2"));
    if (streq(res, "VAL")) return in_list(atom, LIST("CG1", "CG2"));
    return 0;
}
/* residues.rs:163-186 */
static int is_ring_atom(const char *res, const char *atom) {
    if (streq(res, "HIS")) return in_list(atom, LIST("CG", "ND1", "CE1", "NE2", "CD2"));
    if (streq(res, "PHE") || streq(res, "TYR")) return in_list(atom, LIST("CG", "CD1", "CD2", "CE1", "CE2", "CZ"));
    if (streq(res, "TRP")) return in_list(atom, LIST("CG", "CD1", "CD2", "NE1", "CE2", "CE3", "CZ2", "CZ3", "CH2"));
    return 0;
}
/* residues.rs:188-268 */
static int is_sc_plane_atom(const char *res, const char *atom) {
    if (streq(res, "ARG")) return in_list(atom, LIST("NE", "CZ", "NH1", "NH2"));
    if (streq(res, "ASN")) return in_list(atom, LIST("CB", "CG", "OD1", "ND2"));
    if (streq(res, "ASP")) return in_list(atom, LIST("CB", "CG", "OD1", "OD2"));
    if (streq(res, "CYS")) return in_list(atom, LIST("CA", "CB", "SG"));
    if (streq(res, "GLU")) return in_list(atom, LIST("CG", "CD", "OE1", "OE2"));
    if (streq(res, "GLN")) return in_list(atom, LIST("CG", "CD", "OE1", "NE2"));
    if (streq(res, "ILE")) return in_list(atom, LIST("CB", "CG1", "CG2", "CD1"));
    if (streq(res, "LEU")) return in_list(atom, LIST("CB", "CG", "CD1", "CD2"));
    if (streq(res, "LYS")) return in_list(atom, LIST("CG", "CD", "CE", "NZ"));
    if (streq(res, "MET")) return in_list(atom, LIST("CG", "SD", "CE"));
    if (streq(res, "PRO")) return in_list(atom, LIST("N", "CA", "CB", "CG", "CD"));
    if (streq(res, "SER")) return in_list(atom, LIST("CA", "CB", "OG"));
    if (streq(res, "THR")) return in_list(atom, LIST("CA", "CB", "OG1", "CG2"));
    if (streq(res, "VAL")) return in_list(atom, LIST("CA", "CB", "CG1", "CG2"));
    return is_ring_atom(res, atom); /* HIS, PHE, TYR, TRP: same atoms as the ring */
}

uint32_t orc_atom_classes(const OrcAtom *a) {
    uint32_t c = 0;
    if (is_hydrogen_donor(a->resn, a->name)) c |= ORC_CLS_DONOR;
    if (is_hydrogen_acceptor(a->resn, a->name)) c |= ORC_CLS_ACCEPTOR;
    if (is_weak_hydrogen_donor(a)) c |= ORC_CLS_WEAK_DONOR;
    if (is_pos_ionizable(a->resn, a->name)) c |= ORC_CLS_POS;
    if (is_neg_ionizable(a->resn, a->name)) c |= ORC_CLS_NEG;
    if (is_hydrophobic(a->res_resn, a->name)) c |= ORC_CLS_HYDROPHOBIC;
    if (streq(a->res_resn, "CYS") && streq(a->name, "SG")) c |= ORC_CLS_CYS_SG;
    if (streq(a->elem, "H")) c |= ORC_CLS_H;
    if (is_pos_ionizable(a->res_resn, a->name)) c |= ORC_CLS_POS_RESN;
    return c;
}

/* ------------------------------------------------------------------ geometry (pdbtbx Atom::distance/angle/dihedral) */
static double atom_distance(const OrcAtom *a, const OrcAtom *b) {
    double dx = b->x - a->x, dy = b->y - a->y, dz = b->z - a->z;
    return sqrt(dx * dx + dy * dy + dz * dz);
}
static const double RAD2DEG = 180.0 / 3.14159265358979323846264338327950288;

double orc_angle(const double a[3], const double b[3], const double c[3]) {
    double ba[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double bc[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    double abs_ba = sqrt(0.0 + ba[0] * ba[0] + ba[1] * ba[1] + ba[2] * ba[2]);
    double abs_bc = sqrt(0.0 + bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
    double dot = 0.0 + ba[0] * bc[0] + ba[1] * bc[1] + ba[2] * bc[2];
    return acos(dot / (abs_ba * abs_bc)) * RAD2DEG;
}
double orc_dihedral(const double a[3], const double b[3], const double c[3], const double d[3]) {
    double ba[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double bc[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    double cb[3] = {b[0] - c[0], b[1] - c[1], b[2] - c[2]};
    double cd[3] = {d[0] - c[0], d[1] - c[1], d[2] - c[2]};
    double n1[3] = {ba[1] * bc[2] - ba[2] * bc[1], ba[2] * bc[0] - ba[0] * bc[2], ba[0] * bc[1] - ba[1] * bc[0]};
    double n2[3] = {cb[1] * cd[2] - cb[2] * cd[1], cb[2] * cd[0] - cb[0] * cd[2], cb[0] * cd[1] - cb[1] * cd[0]};
    double abs_n1 = sqrt(0.0 + n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    double abs_n2 = sqrt(0.0 + n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
    double dot = 0.0 + n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
    return acos(dot / (abs_n1 * abs_n2)) * RAD2DEG;
}
static void atom_pos(const OrcAtom *a, double p[3]) { p[0] = a->x; p[1] = a->y; p[2] = a->z; }

/* ------------------------------------------------------------------ PDB reader (fixed columns) */
static void field(const char *line, size_t len, int c0, int c1, char *out, int cap, int upper) {
    /* columns are 1-based inclusive; trims blanks */
    int n = 0;
    int a = c0 - 1, b = c1;
    if ((size_t)b > len) b = (int)len;
    while (a < b && isspace((unsigned char)line[a])) a++;
    while (b > a && isspace((unsigned char)line[b - 1])) b--;
    for (int k = a; k < b && n < cap - 1; k++) out[n++] = upper ? (char)toupper((unsigned char)line[k]) : line[k];
    out[n] = 0;
}

static int parse_pdb_records(const char *path, OrcAtom **out, int32_t *n_out) {
    FILE *f = fopen(path, "r");
    if (!f) { set_err("cannot open input file"); return ORC_ERR_IO; }
    int32_t cap = 1024, n = 0;
    OrcAtom *atoms = (OrcAtom *)malloc(sizeof(OrcAtom) * cap);
    char line[256];
    int32_t model_serial = 0;
    while (fgets(line, sizeof line, f)) {
        size_t len = strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (strncmp(line, "MODEL", 5) == 0 && (len == 5 || isspace((unsigned char)line[5]))) {
            char tmp[16]; field(line, len, 7, 14, tmp, sizeof tmp, 0);
            model_serial = (int32_t)strtol(tmp, NULL, 10);
            continue;
        }
        int is_atom = strncmp(line, "ATOM  ", 6) == 0, is_het = strncmp(line, "HETATM", 6) == 0;
        if (!is_atom && !is_het) continue;
        if (len < 54) continue;
        if (n == cap) { cap *= 2; atoms = (OrcAtom *)realloc(atoms, sizeof(OrcAtom) * cap); }
        OrcAtom *a = &atoms[n];
        memset(a, 0, sizeof *a);
        char tmp[32];
        field(line, len, 7, 11, tmp, sizeof tmp, 0); a->serial = (int32_t)strtol(tmp, NULL, 10);
        field(line, len, 13, 16, a->name, sizeof a->name, 1);
        field(line, len, 17, 17, a->altloc, sizeof a->altloc, 0);
        field(line, len, 18, 20, a->resn, sizeof a->resn, 1);
        field(line, len, 22, 22, a->chain, sizeof a->chain, 0);
        field(line, len, 23, 26, tmp, sizeof tmp, 0); a->resi = (int32_t)strtol(tmp, NULL, 10);
        field(line, len, 27, 27, a->icode, sizeof a->icode, 0);
        field(line, len, 31, 38, tmp, sizeof tmp, 0); a->x = strtod(tmp, NULL);
        field(line, len, 39, 46, tmp, sizeof tmp, 0); a->y = strtod(tmp, NULL);
        field(line, len, 47, 54, tmp, sizeof tmp, 0); a->z = strtod(tmp, NULL);
        field(line, len, 55, 60, tmp, sizeof tmp, 0); a->occ = tmp[0] ? strtod(tmp, NULL) : 1.0;
        field(line, len, 77, 78, a->elem, sizeof a->elem, 1);
        if (!a->elem[0]) { /* infer from the atom name: leading letters, digits skipped */
            const char *p = a->name; while (*p && isdigit((unsigned char)*p)) p++;
            a->elem[0] = *p ? *p : 'X'; a->elem[1] = 0;
        }
        a->hetero = is_het;
        a->model_serial = model_serial;
        n++;
    }
    fclose(f);
    *out = atoms; *n_out = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ hierarchy (pdbtbx-like add_atom semantics) */
typedef struct { int32_t chain; int32_t resi; char icode[4]; int32_t first_conf; int32_t n_conf; int keep; char name[8]; int32_t new_idx; int32_t ord; } HRes;
typedef struct { int32_t res; char name[8]; char altloc[4]; int32_t ord; } HConf;

static int cmp_res_atoms(const void *pa, const void *pb, void *ctx) {
    const OrcStructure *s = (const OrcStructure *)ctx;
    int32_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
    if (s->atoms[a].res_idx != s->atoms[b].res_idx) return s->atoms[a].res_idx < s->atoms[b].res_idx ? -1 : 1;
    if (s->conf_ord[a] != s->conf_ord[b]) return s->conf_ord[a] < s->conf_ord[b] ? -1 : 1;
    return a < b ? -1 : (a > b);
}

static int finish_structure(OrcStructure *s) {
    /* residue -> atoms CSR in hierarchy order */
    int32_t n_res = 0;
    for (int32_t i = 0; i < s->n; i++) if (s->atoms[i].res_idx + 1 > n_res) n_res = s->atoms[i].res_idx + 1;
    s->n_res = n_res;
    s->res_ptr = (int32_t *)calloc((size_t)n_res + 1, sizeof(int32_t));
    s->res_atoms = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->n > 0 ? s->n : 1));
    for (int32_t i = 0; i < s->n; i++) { s->res_atoms[i] = i; s->res_ptr[s->atoms[i].res_idx + 1]++; }
    for (int32_t r = 0; r < n_res; r++) s->res_ptr[r + 1] += s->res_ptr[r];
    qsort_r(s->res_atoms, (size_t)s->n, sizeof(int32_t), cmp_res_atoms, s);
    return ORC_OK;
}

OrcStructure *orc_from_atoms(const OrcAtom *in, int32_t n, int flat) {
    OrcStructure *s = (OrcStructure *)calloc(1, sizeof *s);
    if (flat) {
        s->atoms = (OrcAtom *)malloc(sizeof(OrcAtom) * (size_t)(n > 0 ? n : 1));
        memcpy(s->atoms, in, sizeof(OrcAtom) * (size_t)n);
        s->n = n;
        s->conf_ord = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
        /* chain list = distinct (model_idx, chain id) in order of appearance */
        int32_t cap = 16; s->chains = (OrcChain *)malloc(sizeof(OrcChain) * cap);
        for (int32_t i = 0; i < n; i++) {
            int found = 0;
            for (int32_t c = s->n_chains - 1; c >= 0; c--)
                if (s->chains[c].model_idx == in[i].model_idx && streq(s->chains[c].id, in[i].chain)) { found = 1; break; }
            if (!found) {
                if (s->n_chains == cap) { cap *= 2; s->chains = (OrcChain *)realloc(s->chains, sizeof(OrcChain) * cap); }
                OrcChain *c = &s->chains[s->n_chains++];
                c->model_idx = in[i].model_idx; c->model_serial = in[i].model_serial;
                snprintf(c->id, sizeof c->id, "%s", in[i].chain);
            }
        }
        finish_structure(s);
        return s;
    }
    /* pdbtbx: Model::add_atom -> Chain::add_atom -> Residue::add_atom: look up an existing chain by id, an
     * existing residue by (serial, insertion), an existing conformer by (name, altloc); else append. */
    int32_t n_models = 0, chain_cap = 16, res_cap = 256, conf_cap = 256;
    int32_t n_chains = 0, n_res = 0, n_conf = 0;
    OrcChain *chains = (OrcChain *)malloc(sizeof(OrcChain) * chain_cap);
    HRes *res = (HRes *)malloc(sizeof(HRes) * res_cap);
    HConf *conf = (HConf *)malloc(sizeof(HConf) * conf_cap);
    int32_t *a_res = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t *a_conf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t *a_chain = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t *a_model = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t cur_model_serial = 0, cur_model = -1;
    /* last residue / conformer caches keep the common sequential case O(1); the full reverse search is the rule */
    for (int32_t i = 0; i < n; i++) {
        const OrcAtom *a = &in[i];
        if (cur_model < 0 || a->model_serial != cur_model_serial) { cur_model = n_models++; cur_model_serial = a->model_serial; }
        int32_t ci = -1;
        for (int32_t c = n_chains - 1; c >= 0 && chains[c].model_idx == cur_model; c--)
            if (streq(chains[c].id, a->chain)) { ci = c; break; }
        if (ci < 0) {
            if (n_chains == chain_cap) { chain_cap *= 2; chains = (OrcChain *)realloc(chains, sizeof(OrcChain) * chain_cap); }
            ci = n_chains++;
            chains[ci].model_idx = cur_model; chains[ci].model_serial = cur_model_serial;
            snprintf(chains[ci].id, sizeof chains[ci].id, "%s", a->chain);
        }
        int32_t ri = -1;
        for (int32_t r = n_res - 1; r >= 0; r--) {
            if (chains[res[r].chain].model_idx != cur_model) break;
            if (res[r].chain == ci && res[r].resi == a->resi && streq(res[r].icode, a->icode)) { ri = r; break; }
        }
        if (ri < 0) {
            if (n_res == res_cap) { res_cap *= 2; res = (HRes *)realloc(res, sizeof(HRes) * res_cap); }
            ri = n_res++;
            memset(&res[ri], 0, sizeof(HRes));
            res[ri].chain = ci; res[ri].resi = a->resi; snprintf(res[ri].icode, sizeof res[ri].icode, "%s", a->icode);
            res[ri].first_conf = -1;
        }
        int32_t ki = -1;
        for (int32_t k = n_conf - 1; k >= 0 && res[ri].first_conf >= 0 && k >= res[ri].first_conf; k--)
            if (conf[k].res == ri && streq(conf[k].name, a->resn) && streq(conf[k].altloc, a->altloc)) { ki = k; break; }
        if (ki < 0) {
            if (n_conf == conf_cap) { conf_cap *= 2; conf = (HConf *)realloc(conf, sizeof(HConf) * conf_cap); }
            ki = n_conf++;
            conf[ki].res = ri; snprintf(conf[ki].name, sizeof conf[ki].name, "%s", a->resn);
            snprintf(conf[ki].altloc, sizeof conf[ki].altloc, "%s", a->altloc);
            conf[ki].ord = res[ri].n_conf++;
            if (res[ri].first_conf < 0) res[ri].first_conf = ki;
        }
        a_res[i] = ri; a_conf[i] = ki; a_chain[i] = ci; a_model[i] = cur_model;
    }
    /* Residue::name(): Some(name) iff every conformer has the same name */
    for (int32_t r = 0; r < n_res; r++) { snprintf(res[r].name, sizeof res[r].name, "%s", conf[res[r].first_conf].name); res[r].keep = 1; }
    int bad_name = 0;
    for (int32_t k = 0; k < n_conf; k++) if (!streq(conf[k].name, res[conf[k].res].name)) { res[conf[k].res].name[0] = 0; bad_name = 1; }
    /* load_model: pdb.remove_residues_by(|res| res.resn().is_none()); resn() unwraps name() -> the reference
     * panics on a residue whose conformers disagree on the name. */
    int rc = ORC_OK;
    if (bad_name) { set_err("residue with conformers of different names (reference panics in load_model)"); rc = ORC_ERR_BAD_INPUT; }
    /* positional index of each surviving residue within its chain (complex.rs:411-440) */
    int32_t *chain_count = (int32_t *)calloc((size_t)(n_chains > 0 ? n_chains : 1), sizeof(int32_t));
    int32_t kept_res = 0;
    for (int32_t r = 0; r < n_res && rc == ORC_OK; r++) {
        char up[8]; size_t L = strlen(res[r].name);
        for (size_t q = 0; q <= L; q++) up[q] = (char)toupper((unsigned char)res[r].name[q]);
        res[r].keep = is_known_residue(up);
        if (res[r].keep) { res[r].ord = chain_count[res[r].chain]++; res[r].new_idx = kept_res++; }
    }
    if (rc == ORC_OK) {
        s->atoms = (OrcAtom *)malloc(sizeof(OrcAtom) * (size_t)(n > 0 ? n : 1));
        s->conf_ord = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        int32_t m = 0;
        for (int32_t i = 0; i < n; i++) {
            const HRes *r = &res[a_res[i]];
            if (!r->keep) continue;
            OrcAtom *o = &s->atoms[m];
            *o = in[i];
            o->model_idx = a_model[i]; o->chain_idx = a_chain[i]; o->res_idx = r->new_idx; o->res_ord = r->ord;
            snprintf(o->res_resn, sizeof o->res_resn, "%s", r->name);
            s->conf_ord[m] = conf[a_conf[i]].ord;
            m++;
        }
        s->n = m;
        s->chains = chains; s->n_chains = n_chains; chains = NULL;
        finish_structure(s);
    }
    free(chains); free(res); free(conf); free(a_res); free(a_conf); free(a_chain); free(a_model); free(chain_count);
    if (rc != ORC_OK) { orc_free_structure(s); return NULL; }
    return s;
}

OrcStructure *orc_load_model(const char *path, int ignore_zero_occupancy) {
    OrcAtom *recs = NULL; int32_t n = 0;
    if (parse_pdb_records(path, &recs, &n) != ORC_OK) return NULL;
    OrcStructure *s = orc_from_atoms(recs, n, 0);
    free(recs);
    if (s && ignore_zero_occupancy) {
        /* python.rs:45-47: pdb.remove_atoms_by(|atom| atom.occupancy() == 0.0); residues keep their ordinals */
        int32_t m = 0;
        for (int32_t i = 0; i < s->n; i++)
            if (!(s->atoms[i].occ == 0.0)) { s->atoms[m] = s->atoms[i]; s->conf_ord[m] = s->conf_ord[i]; m++; }
        s->n = m;
        free(s->res_ptr); free(s->res_atoms); s->res_ptr = NULL; s->res_atoms = NULL;
        finish_structure(s);
    }
    return s;
}

/* ------------------------------------------------------------------ parse_groups (utils.rs:71-115) */
typedef struct { char (*ids)[8]; int n; int sorted; } ChainSet;

static int cmp_id(const void *a, const void *b) { return strcmp((const char *)a, (const char *)b); }
static int set_has(const ChainSet *s, const char *id) {
    if (s->sorted) return bsearch(id, s->ids, (size_t)s->n, sizeof(char[8]), cmp_id) != NULL; /* many-chain inputs */
    for (int i = 0; i < s->n; i++) if (streq(s->ids[i], id)) return 1;
    return 0;
}
static void set_add(ChainSet *s, const char *id) {
    if (set_has(s, id)) return;
    s->ids = (char(*)[8])realloc(s->ids, sizeof(char[8]) * (size_t)(s->n + 1));
    snprintf(s->ids[s->n++], 8, "%s", id);
}
static void split_commas(const char *p, size_t len, ChainSet *out) {
    size_t a = 0;
    for (size_t k = 0; k <= len; k++) {
        if (k == len || p[k] == ',') {
            if (k > a) { char id[8]; size_t L = k - a; if (L > 7) L = 7; memcpy(id, p + a, L); id[L] = 0; set_add(out, id); }
            a = k + 1;
        }
    }
}
static int parse_groups_sets(const ChainSet *all, const char *groups, ChainSet *lig, ChainSet *rec) {
    const char *slash = strchr(groups, '/');
    if (!slash) { set_err("Invalid chain groups format! Use '/' for all-to-all comparisons."); return ORC_ERR_BAD_GROUPS; }
    const char *second = slash + 1;
    const char *slash2 = strchr(second, '/');
    size_t len2 = slash2 ? (size_t)(slash2 - second) : strlen(second);
    split_commas(groups, (size_t)(slash - groups), lig);
    split_commas(second, len2, rec);
    if (lig->n == 0 && rec->n == 0) {
        for (int i = 0; i < all->n; i++) { set_add(lig, all->ids[i]); set_add(rec, all->ids[i]); }
        return ORC_OK;
    }
    if (lig->n == 0) { for (int i = 0; i < all->n; i++) if (!set_has(rec, all->ids[i])) set_add(lig, all->ids[i]); }
    else if (rec->n == 0) { for (int i = 0; i < all->n; i++) if (!set_has(lig, all->ids[i])) set_add(rec, all->ids[i]); }
    if (lig->n == 0 || rec->n == 0) { set_err("Empty chain groups!"); return ORC_ERR_EMPTY_GROUPS; }
    return ORC_OK;
}
static void emit_set(ChainSet *s, char *out, int cap, int *n) {
    qsort(s->ids, (size_t)s->n, sizeof(char[8]), cmp_id);
    int pos = 0; *n = 0;
    for (int i = 0; i < s->n; i++) {
        int L = (int)strlen(s->ids[i]) + 1;
        if (pos + L > cap) break;
        memcpy(out + pos, s->ids[i], (size_t)L); pos += L; (*n)++;
    }
}
int orc_parse_groups(const char *const *all_chains, int n_chains, const char *groups, char *lig_out, int lig_cap,
                     int *n_lig, char *rec_out, int rec_cap, int *n_rec) {
    ChainSet all = {0}, lig = {0}, rec = {0};
    for (int i = 0; i < n_chains; i++) set_add(&all, all_chains[i]);
    int rc = parse_groups_sets(&all, groups, &lig, &rec);
    if (rc == ORC_OK) { emit_set(&lig, lig_out, lig_cap, n_lig); emit_set(&rec, rec_out, rec_cap, n_rec); }
    free(all.ids); free(lig.ids); free(rec.ids);
    return rc;
}
static int structure_groups(const OrcStructure *s, const char *groups, ChainSet *lig, ChainSet *rec) {
    ChainSet all = {0};
    for (int32_t c = 0; c < s->n_chains; c++) set_add(&all, s->chains[c].id);
    int rc = parse_groups_sets(&all, groups, lig, rec);
    free(all.ids);
    if (rc == ORC_OK) {
        qsort(lig->ids, (size_t)lig->n, sizeof(char[8]), cmp_id); lig->sorted = 1;
        qsort(rec->ids, (size_t)rec->n, sizeof(char[8]), cmp_id); rec->sorted = 1;
    }
    return rc;
}

/* ------------------------------------------------------------------ pair filter (complex.rs:76-131) */
typedef struct {
    const OrcStructure *s;
    ChainSet lig, rec;
    double vdw_comp, cutoff;
} Complex;

typedef struct { int32_t model_serial; const char *chain; int32_t res_ord; } ResKey;

static int should_compare_residues(const Complex *cx, const ResKey *r1, const ResKey *r2, int symmetric) {
    if (r1->model_serial != r2->model_serial) return 0;
    int l1 = set_has(&cx->lig, r1->chain), l2 = set_has(&cx->lig, r2->chain);
    int c1 = set_has(&cx->rec, r1->chain), c2 = set_has(&cx->rec, r2->chain);
    if (!((l1 && c2) | (l2 && c1))) return 0;
    if (streq(r1->chain, r2->chain)) {
        int64_t e1 = r1->res_ord, e2 = r2->res_ord;
        if (symmetric) return (e2 > 1) && (e1 < e2 - 1);
        int neigh = (e1 == 0) ? ((e2 == e1) | (e2 == e1 + 1)) : ((e2 == e1 - 1) | (e2 == e1) | (e2 == e1 + 1));
        return !neigh;
    }
    return !(symmetric && c1 && c2 && l1 && l2 && (strcmp(r1->chain, r2->chain) > 0));
}
static int should_compare_entities(const Complex *cx, const OrcAtom *e1, const OrcAtom *e2, int symmetric) {
    if (streq(e1->elem, "H") | streq(e2->elem, "H")) return 0;
    ResKey r1 = {e1->model_serial, e1->chain, e1->res_ord}, r2 = {e2->model_serial, e2->chain, e2->res_ord};
    return should_compare_residues(cx, &r1, &r2, symmetric);
}

/* ------------------------------------------------------------------ rules */
/* vdw.rs:46-80 */
static int is_disulfide(const OrcStructure *s, const OrcAtom *e1, const OrcAtom *e2) {
    if (!(streq(e1->res_resn, "CYS") && streq(e2->res_resn, "CYS") && streq(e1->name, "SG") && streq(e2->name, "SG"))) return 0;
    const OrcAtom *q[4] = {NULL, NULL, NULL, NULL}; /* cb1 s1 s2 cb2 */
    const OrcAtom *es[2] = {e1, e2};
    for (int k = 0; k < 2; k++) {
        int32_t r = es[k]->res_idx;
        for (int32_t p = s->res_ptr[r]; p < s->res_ptr[r + 1]; p++) {
            const OrcAtom *a = &s->atoms[s->res_atoms[p]];
            if (!q[k == 0 ? 0 : 3] && streq(a->name, "CB")) q[k == 0 ? 0 : 3] = a;
            if (!q[k == 0 ? 1 : 2] && streq(a->name, "SG")) q[k == 0 ? 1 : 2] = a;
        }
    }
    if (!q[0] || !q[3]) return -1; /* reference unwrap() panics */
    double a[3], b[3], c[3], d[3];
    atom_pos(q[0], a); atom_pos(q[1], b); atom_pos(q[2], c); atom_pos(q[3], d);
    double dih = fabs(orc_dihedral(a, b, c, d));
    return dih >= 60.0 && dih <= 120.0;
}

static int hbond_like(const OrcStructure *s, const OrcAtom *donor, const OrcAtom *acceptor, double c, double min_angle,
                      int strong_code, int polar_code) {
    double da = atom_distance(donor, acceptor);
    if (da <= 4.0) {
        double cov, acc_vdw, h_cov, h_vdw;
        orc_radii(acceptor->elem, &cov, &acc_vdw);
        orc_radii("H", &h_cov, &h_vdw);
        int32_t r = donor->res_idx;
        for (int32_t p = s->res_ptr[r]; p < s->res_ptr[r + 1]; p++) {
            const OrcAtom *h = &s->atoms[s->res_atoms[p]];
            if (!streq(h->elem, "H")) continue;
            double pd[3], ph[3], pa[3];
            atom_pos(donor, pd); atom_pos(h, ph); atom_pos(acceptor, pa);
            if ((atom_distance(h, acceptor) <= h_vdw + acc_vdw + c) && (orc_angle(pd, ph, pa) >= min_angle)) return strong_code;
        }
    }
    if (da <= 3.5) return polar_code;
    return -1;
}
/* hbond.rs:30-66 */
static int find_hydrogen_bond(const OrcStructure *s, const OrcAtom *e1, const OrcAtom *e2, double c) {
    const OrcAtom *d = NULL, *a = NULL;
    if (is_hydrogen_donor(e1->resn, e1->name) && is_hydrogen_acceptor(e2->resn, e2->name)) { d = e1; a = e2; }
    else if (is_hydrogen_donor(e2->resn, e2->name) && is_hydrogen_acceptor(e1->resn, e1->name)) { d = e2; a = e1; }
    else return -1;
    return hbond_like(s, d, a, c, 90.0, ORC_HydrogenBond, ORC_PolarContact);
}
/* hbond.rs:74-110 */
static int find_weak_hydrogen_bond(const OrcStructure *s, const OrcAtom *e1, const OrcAtom *e2, double c) {
    const OrcAtom *d = NULL, *a = NULL;
    if (is_weak_hydrogen_donor(e1) && is_hydrogen_acceptor(e2->resn, e2->name)) { d = e1; a = e2; }
    else if (is_weak_hydrogen_donor(e2) && is_hydrogen_acceptor(e1->resn, e1->name)) { d = e2; a = e1; }
    else return -1;
    return hbond_like(s, d, a, c, 130.0, ORC_WeakHydrogenBond, ORC_WeakPolarContact);
}

/* complex.rs:217-296: all rows of one candidate pair as a bit set; -1 on a reference panic */
static int64_t classify_pair(const OrcStructure *s, const OrcAtom *e1, const OrcAtom *e2, double c, double *dist_out) {
    uint32_t kind = 0;
    double dist = atom_distance(e1, e2);
    *dist_out = dist;
    double cov1, vdw1, cov2, vdw2;
    if (!orc_radii(e1->elem, &cov1, &vdw1) || !orc_radii(e2->elem, &cov2, &vdw2)) { set_err("element without radii"); return -1; }
    double sum_cov = cov1 + cov2, sum_vdw = vdw1 + vdw2;
    /* vdw.rs:32-43 */
    if (dist < sum_cov - c) { kind |= 1u << ORC_StericClash; return kind; } /* complex.rs:233-235 */
    else if (dist < sum_cov + c) {
        int ds = is_disulfide(s, e1, e2);
        if (ds < 0) { set_err("CYS without CB (reference panics in is_disulfide)"); return -1; }
        kind |= 1u << (ds ? ORC_Disulfide : ORC_CovalentBond);
    } else if (dist < sum_vdw + c) kind |= 1u << ORC_VanDerWaalsContact;
    /* ionic.rs:11-22 */
    int ionic = -1;
    {
        int pair = (is_pos_ionizable(e1->resn, e1->name) && is_neg_ionizable(e2->resn, e2->name)) ||
                   (is_pos_ionizable(e2->resn, e2->name) & is_neg_ionizable(e1->resn, e1->name));
        if (pair && dist <= 4.0) ionic = ORC_IonicBond;
    }
    int hb = find_hydrogen_bond(s, e1, e2, c);
    /* complex.rs:240-251 */
    int electro = -1;
    if (ionic >= 0 && hb >= 0) electro = (hb == ORC_HydrogenBond) ? ORC_SaltBridge : ionic;
    else if (ionic >= 0) electro = ionic;
    else if (hb >= 0) electro = hb;
    if (electro >= 0) kind |= 1u << electro;
    int weak = find_weak_hydrogen_bond(s, e1, e2, c);
    if (weak >= 0) kind |= 1u << weak;
    /* ionic.rs:25-35,59-81 */
    {
        int both_pos = is_pos_ionizable(e1->resn, e1->name) && is_pos_ionizable(e2->resn, e2->name);
        int both_neg = is_neg_ionizable(e1->resn, e1->name) && is_neg_ionizable(e2->resn, e2->name);
        if ((both_pos | both_neg) && dist <= 4.0) kind |= 1u << ORC_IonicRepulsion;
    }
    /* hydrophobic.rs:10-24 */
    if (is_hydrophobic(e1->res_resn, e1->name) && is_hydrophobic(e2->res_resn, e2->name) && dist <= 4.5)
        kind |= 1u << ORC_HydrophobicContact;
    return kind;
}

/* ------------------------------------------------------------------ spatial index (stands in for rstar) */
typedef struct { int64_t key; int32_t idx; } CellEnt;
typedef struct {
    CellEnt *ents; int32_t n;
    double ox, oy, oz, edge;
} Grid;
static int cmp_cell(const void *a, const void *b) {
    const CellEnt *x = (const CellEnt *)a, *y = (const CellEnt *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
#define CELL_BITS 21
static int64_t cell_key(int64_t cx, int64_t cy, int64_t cz) { return (cz << (2 * CELL_BITS)) | (cy << CELL_BITS) | cx; }
static void grid_build(Grid *g, const OrcAtom *atoms, int32_t n, double edge) {
    g->n = n; g->edge = edge > 1e-6 ? edge * 1.000001 : 1e-6; /* a hair above the cutoff: |dx| <= cutoff never skips a cell after rounding */
    g->ox = g->oy = g->oz = 0.0;
    for (int32_t i = 0; i < n; i++) {
        if (i == 0 || atoms[i].x < g->ox) g->ox = atoms[i].x;
        if (i == 0 || atoms[i].y < g->oy) g->oy = atoms[i].y;
        if (i == 0 || atoms[i].z < g->oz) g->oz = atoms[i].z;
    }
    g->ents = (CellEnt *)malloc(sizeof(CellEnt) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; i++) {
        int64_t cx = (int64_t)floor((atoms[i].x - g->ox) / g->edge) + 1, cy = (int64_t)floor((atoms[i].y - g->oy) / g->edge) + 1,
                cz = (int64_t)floor((atoms[i].z - g->oz) / g->edge) + 1;
        g->ents[i].key = cell_key(cx, cy, cz); g->ents[i].idx = i;
    }
    qsort(g->ents, (size_t)n, sizeof(CellEnt), cmp_cell);
}
static int32_t grid_lower(const Grid *g, int64_t key) {
    int32_t lo = 0, hi = g->n;
    while (lo < hi) { int32_t mid = lo + (hi - lo) / 2; if (g->ents[mid].key < key) lo = mid + 1; else hi = mid; }
    return lo;
}

typedef struct { OrcPair *p; int64_t n, cap; } PairVec;
static void pv_push(PairVec *v, int32_t i, int32_t j, double d, uint32_t kind) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 4096; v->p = (OrcPair *)realloc(v->p, sizeof(OrcPair) * (size_t)v->cap); }
    OrcPair *q = &v->p[v->n++]; q->i = i; q->j = j; q->dist = d; q->kind = kind; q->pad = 0;
}

static int complex_init(Complex *cx, const OrcStructure *s, const char *groups, double vdw_comp, double cutoff) {
    memset(cx, 0, sizeof *cx);
    cx->s = s; cx->vdw_comp = vdw_comp; cx->cutoff = cutoff;
    return structure_groups(s, groups, &cx->lig, &cx->rec);
}
static void complex_free(Complex *cx) { free(cx->lig.ids); free(cx->rec.ids); }

/* complex.rs:189-299 */
static int atomic_contacts(const Complex *cx, int mode, PairVec *out) {
    const OrcStructure *s = cx->s;
    const OrcAtom *A = s->atoms;
    const double r2 = cx->cutoff * cx->cutoff;
    Grid g = {0};
    if (mode == 0) grid_build(&g, A, s->n, cx->cutoff);
    int rc = ORC_OK;
    for (int32_t xi = 0; xi < s->n && rc == ORC_OK; xi++) {
        const OrcAtom *x = &A[xi];
        if (!(set_has(&cx->lig, x->chain) && !streq(x->elem, "H"))) continue; /* :200-202 */
#define VISIT(yi_)                                                                                          \
    do {                                                                                                    \
        const OrcAtom *y = &A[(yi_)];                                                                       \
        double dx = y->x - x->x, dy = y->y - x->y, dz = y->z - x->z;                                        \
        double d2 = dx * dx + dy * dy + dz * dz;                                                            \
        if (d2 <= r2 && set_has(&cx->rec, y->chain) && should_compare_entities(cx, x, y, 1)) {              \
            double dist; int64_t k = classify_pair(s, x, y, cx->vdw_comp, &dist);                           \
            if (k < 0) { rc = ORC_ERR_BAD_INPUT; break; }                                                   \
            pv_push(out, xi, (yi_), dist, (uint32_t)k);                                                     \
        }                                                                                                   \
    } while (0)
        if (mode == 1) {
            for (int32_t yi = 0; yi < s->n; yi++) VISIT(yi);
        } else {
            int64_t cx0 = (int64_t)floor((x->x - g.ox) / g.edge) + 1, cy0 = (int64_t)floor((x->y - g.oy) / g.edge) + 1,
                    cz0 = (int64_t)floor((x->z - g.oz) / g.edge) + 1;
            for (int64_t dz_ = -1; dz_ <= 1 && rc == ORC_OK; dz_++)
                for (int64_t dy_ = -1; dy_ <= 1 && rc == ORC_OK; dy_++) {
                    /* the three x-adjacent cells are one contiguous key range */
                    int64_t k0 = cell_key(cx0 - 1, cy0 + dy_, cz0 + dz_), k1 = cell_key(cx0 + 1, cy0 + dy_, cz0 + dz_);
                    for (int32_t p = grid_lower(&g, k0); p < g.n && g.ents[p].key <= k1; p++) VISIT(g.ents[p].idx);
                }
        }
#undef VISIT
    }
    free(g.ents);
    return rc;
}

int orc_atomic_contacts(const OrcStructure *s, const char *groups, double vdw_comp, double dist_cutoff, int mode,
                        OrcPair **pairs, int64_t *n_pairs) {
    Complex cx;
    int rc = complex_init(&cx, s, groups, vdw_comp, dist_cutoff);
    PairVec v = {0};
    if (rc == ORC_OK) rc = atomic_contacts(&cx, mode, &v);
    complex_free(&cx);
    if (rc != ORC_OK) { free(v.p); *pairs = NULL; *n_pairs = 0; return rc; }
    *pairs = v.p; *n_pairs = v.n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ planes (residues.rs:24-75,270-298) */
/* One-sided (Hestenes) Jacobi SVD of the centred 3xN coordinate matrix A: rotate the three ROWS of A until they
 * are mutually orthogonal; the accumulated rotation is U, the row norms are the singular values.  normal =
 * column of U with the smallest singular value (nalgebra: svd.u.column(2), singular values sorted descending). */
static int plane_fit(const double (*pts)[3], int n, double center[3], double normal[3]) {
    if (n < 3) return 0;
    double c[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) { c[0] += pts[i][0]; c[1] += pts[i][1]; c[2] += pts[i][2]; }
    c[0] /= n; c[1] /= n; c[2] /= n;
    double *R = (double *)malloc(sizeof(double) * 3 * (size_t)n); /* rows */
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) R[k * n + i] = pts[i][k] - c[k];
    double U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < n; i++) { alpha += R[p * n + i] * R[p * n + i]; beta += R[q * n + i] * R[q * n + i]; gamma += R[p * n + i] * R[q * n + i]; }
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                off += fabs(gamma);
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; i++) {
                    double a = R[p * n + i], b = R[q * n + i];
                    R[p * n + i] = cs * a - sn * b; R[q * n + i] = sn * a + cs * b;
                }
                for (int k = 0; k < 3; k++) { double a = U[k][p], b = U[k][q]; U[k][p] = cs * a - sn * b; U[k][q] = sn * a + cs * b; }
            }
        if (off == 0.0) break;
    }
    int best = 0; double bestv = 0;
    for (int k = 0; k < 3; k++) {
        double nv = 0; for (int i = 0; i < n; i++) nv += R[k * n + i] * R[k * n + i];
        if (k == 0 || nv < bestv) { best = k; bestv = nv; }
    }
    double nn = sqrt(U[0][best] * U[0][best] + U[1][best] * U[1][best] + U[2][best] * U[2][best]);
    for (int k = 0; k < 3; k++) { center[k] = c[k]; normal[k] = U[k][best] / nn; }
    free(R);
    return 1;
}
static double plane_point_dist(const OrcPlane *p, const double q[3]) {
    double v[3] = {q[0] - p->c[0], q[1] - p->c[1], q[2] - p->c[2]};
    return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}
static double fold_deg(double rad) {
    if (rad > 1.57079632679489661923) rad = 3.14159265358979323846 - rad;
    return rad * RAD2DEG;
}
static double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static double plane_point_angle(const OrcPlane *p, const double q[3]) {
    double v[3] = {q[0] - p->c[0], q[1] - p->c[1], q[2] - p->c[2]};
    double dot = p->n[0] * v[0] + p->n[1] * v[1] + p->n[2] * v[2];
    return fold_deg(acos(dot / (norm3(p->n) * norm3(v))));
}
static double plane_dihedral(const OrcPlane *a, const OrcPlane *b) {
    double dot = a->n[0] * b->n[0] + a->n[1] * b->n[1] + a->n[2] * b->n[2];
    return fold_deg(acos(dot / (norm3(a->n) * norm3(b->n))));
}

/* residues.rs:31-75 exposed for the plane-identity tests (residues.rs:306-332): out = {|c2-c1|, dihedral, angle(p1, c2)} */
void orc_plane_metrics(const double c1[3], const double n1[3], const double c2[3], const double n2[3], double out[3]) {
    OrcPlane a, b;
    memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
    memcpy(a.c, c1, sizeof a.c); memcpy(a.n, n1, sizeof a.n); memcpy(b.c, c2, sizeof b.c); memcpy(b.n, n2, sizeof b.n);
    out[0] = plane_point_dist(&a, b.c); out[1] = plane_dihedral(&a, &b); out[2] = plane_point_angle(&a, b.c);
}

/* complex.rs:442-514 (quirk: for every model serial, ALL chains of ALL models are visited; a later model's
 * residue with the same id overwrites the earlier plane).  One entry per (model serial, conformer key). */
typedef struct { OrcPlane *p; int32_t n, cap; } PlaneVec;
static int same_plane_key(const OrcPlane *a, int32_t model_serial, const OrcAtom *r, const char *altloc) {
    return a->model_serial == model_serial && streq(a->chain, r->chain) && a->resi == r->resi &&
           streq(a->insertion, r->icode) && streq(a->altloc, altloc) && streq(a->resn, r->res_resn);
}
static int build_planes(const OrcStructure *s, int which, PlaneVec *out, int *n_failed) {
    /* distinct model serials in order */
    int32_t nm = 0; int32_t *serials = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->n_chains + 1));
    for (int32_t c = 0; c < s->n_chains; c++) {
        int f = 0; for (int32_t k = 0; k < nm; k++) if (serials[k] == s->chains[c].model_serial) f = 1;
        if (!f) serials[nm++] = s->chains[c].model_serial;
    }
    *n_failed = 0;
    for (int32_t m = 0; m < nm; m++) {
        for (int32_t r = 0; r < s->n_res; r++) {
            int32_t p0 = s->res_ptr[r], p1 = s->res_ptr[r + 1];
            if (p0 == p1) continue;
            const OrcAtom *first = &s->atoms[s->res_atoms[p0]];
            const char *resn = first->res_resn;
            if (which == 0 && !in_list(resn, LIST("HIS", "PHE", "TYR", "TRP"))) continue;
            int cnt = 0;
            double (*pts)[3] = (double (*)[3])malloc(sizeof(double[3]) * (size_t)(p1 - p0));
            for (int32_t p = p0; p < p1; p++) {
                const OrcAtom *a = &s->atoms[s->res_atoms[p]];
                if (which == 0 ? is_ring_atom(resn, a->name) : is_sc_plane_atom(resn, a->name)) { atom_pos(a, pts[cnt]); cnt++; }
            }
            double c[3], nrm[3];
            int ok = plane_fit(pts, cnt, c, nrm);
            free(pts);
            /* one key per conformer (distinct altloc strings of the residue's atoms) */
            for (int32_t p = p0; p < p1; p++) {
                const OrcAtom *a = &s->atoms[s->res_atoms[p]];
                int dup = 0;
                for (int32_t q = p0; q < p; q++) if (streq(s->atoms[s->res_atoms[q]].altloc, a->altloc)) { dup = 1; break; }
                if (dup) continue;
                if (!ok) { if (which == 0) (*n_failed)++; continue; }
                OrcPlane *dst = NULL;
                for (int32_t k = 0; k < out->n; k++) if (same_plane_key(&out->p[k], serials[m], first, a->altloc)) { dst = &out->p[k]; break; }
                if (!dst) {
                    if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 64; out->p = (OrcPlane *)realloc(out->p, sizeof(OrcPlane) * (size_t)out->cap); }
                    dst = &out->p[out->n++];
                    memset(dst, 0, sizeof *dst);
                }
                memcpy(dst->c, c, sizeof c); memcpy(dst->n, nrm, sizeof nrm);
                dst->model_serial = serials[m]; dst->resi = first->resi; dst->res_idx = r;
                /* res_ord used by the pair filter comes from res2idx[(model serial, chain, resi, ...)] */
                dst->res_ord = first->res_ord;
                snprintf(dst->chain, sizeof dst->chain, "%s", first->chain);
                snprintf(dst->resn, sizeof dst->resn, "%s", resn);
                snprintf(dst->insertion, sizeof dst->insertion, "%s", first->icode);
                snprintf(dst->altloc, sizeof dst->altloc, "%s", a->altloc);
            }
        }
    }
    free(serials);
    return ORC_OK;
}
/* res2idx lookup for a plane key that was inserted under model serial m but may come from another model's chain:
 * find the residue ordinal of (m, chain, resi, icode, altloc, resn) among the atoms of model serial m. */
static int plane_res_ord(const OrcStructure *s, const OrcPlane *p, int32_t *ord) {
    for (int32_t i = 0; i < s->n; i++) {
        const OrcAtom *a = &s->atoms[i];
        if (a->model_serial == p->model_serial && streq(a->chain, p->chain) && a->resi == p->resi && streq(a->icode, p->insertion) &&
            streq(a->altloc, p->altloc) && streq(a->res_resn, p->resn)) { *ord = a->res_ord; return 1; }
    }
    return 0;
}

int orc_planes(const OrcStructure *s, int which, OrcPlane **planes, int32_t *n_planes) {
    PlaneVec v = {0}; int nf = 0;
    build_planes(s, which, &v, &nf);
    *planes = v.p; *n_planes = v.n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ table */
typedef struct { OrcRow *r; int64_t n, cap; } RowVec;
static OrcRow *rv_push(RowVec *v) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->r = (OrcRow *)realloc(v->r, sizeof(OrcRow) * (size_t)v->cap); }
    OrcRow *r = &v->r[v->n++]; memset(r, 0, sizeof *r); return r;
}
/* structs.rs:109-119 InteractingEntity::from_hier */
static void entity_from_atom(OrcEntity *e, const OrcAtom *a) {
    snprintf(e->chain, sizeof e->chain, "%s", a->chain);
    snprintf(e->resn, sizeof e->resn, "%s", a->res_resn);
    snprintf(e->insertion, sizeof e->insertion, "%s", a->icode);
    snprintf(e->altloc, sizeof e->altloc, "%s", a->altloc);
    snprintf(e->atomn, sizeof e->atomn, "%s", a->name);
    e->resi = a->resi; e->atomi = a->serial;
}
static void entity_from_ring(OrcEntity *e, const OrcPlane *p) {
    snprintf(e->chain, sizeof e->chain, "%s", p->chain);
    snprintf(e->resn, sizeof e->resn, "%s", p->resn);
    snprintf(e->insertion, sizeof e->insertion, "%s", p->insertion);
    snprintf(e->altloc, sizeof e->altloc, "%s", p->altloc);
    snprintf(e->atomn, sizeof e->atomn, "Ring");
    e->resi = p->resi; e->atomi = 0;
}
static const OrcPlane *find_plane(const PlaneVec *v, uint32_t model, const OrcEntity *e) {
    for (int32_t k = 0; k < v->n; k++) {
        const OrcPlane *p = &v->p[k];
        if ((uint32_t)p->model_serial == model && streq(p->chain, e->chain) && p->resi == e->resi && streq(p->insertion, e->insertion) &&
            streq(p->altloc, e->altloc) && streq(p->resn, e->resn)) return p;
    }
    return NULL;
}
/* mod.rs:120-134 */
static int cmp_rows(const void *pa, const void *pb) {
    const OrcRow *a = (const OrcRow *)pa, *b = (const OrcRow *)pb;
    int c;
    if (a->model != b->model) return a->model < b->model ? -1 : 1;
    if ((c = strcmp(a->from.chain, b->from.chain))) return c;
    if ((c = strcmp(a->to.chain, b->to.chain))) return c;
    if (a->from.resi != b->from.resi) return a->from.resi < b->from.resi ? -1 : 1;
    if ((c = strcmp(a->from.altloc, b->from.altloc))) return c;
    if (a->from.atomi != b->from.atomi) return a->from.atomi < b->from.atomi ? -1 : 1;
    if (a->to.resi != b->to.resi) return a->to.resi < b->to.resi ? -1 : 1;
    if ((c = strcmp(a->to.altloc, b->to.altloc))) return c;
    if (a->to.atomi != b->to.atomi) return a->to.atomi < b->to.atomi ? -1 : 1;
    if ((c = strcmp(orc_interaction_name(a->interaction), orc_interaction_name(b->interaction)))) return c;
    /* the reference sort is unstable on full ties; break them deterministically */
    if ((c = strcmp(a->from.insertion, b->from.insertion))) return c;
    if ((c = strcmp(a->to.insertion, b->to.insertion))) return c;
    if (a->distance != b->distance) return a->distance < b->distance ? -1 : 1;
    return 0;
}

int orc_get_contacts(const OrcStructure *s, const char *groups, double vdw_comp, double dist_cutoff, OrcRow **rows_out,
                     int64_t *n_rows) {
    *rows_out = NULL; *n_rows = 0;
    Complex cx;
    int rc = complex_init(&cx, s, groups, vdw_comp, dist_cutoff);
    if (rc != ORC_OK) { complex_free(&cx); return rc; }
    PlaneVec rings = {0}, scp = {0}; int nf = 0, nf2 = 0;
    build_planes(s, 0, &rings, &nf);
    if (rings.n == 0) { set_err("Error building ring positions"); complex_free(&cx); free(rings.p); return ORC_ERR_NO_RINGS; } /* complex.rs:50 */
    build_planes(s, 1, &scp, &nf2);
    RowVec rows = {0};
    /* atom-atom rows */
    PairVec pv = {0};
    rc = atomic_contacts(&cx, 0, &pv);
    if (rc == ORC_OK) {
        for (int64_t k = 0; k < pv.n; k++)
            for (int b = 0; b < ORC_N_INTERACTIONS; b++)
                if (pv.p[k].kind & (1u << b)) {
                    OrcRow *r = rv_push(&rows);
                    const OrcAtom *x = &s->atoms[pv.p[k].i], *y = &s->atoms[pv.p[k].j];
                    r->model = (uint32_t)x->model_serial; r->interaction = b; r->distance = pv.p[k].dist;
                    entity_from_atom(&r->from, x); entity_from_atom(&r->to, y);
                    r->from_atom = pv.p[k].i; r->to_atom = pv.p[k].j;
                }
        /* ring-atom rows (complex.rs:301-352, aromatic.rs:14-29) */
        const double r2 = dist_cutoff * dist_cutoff;
        int32_t *ring_ord_v = (int32_t *)malloc(sizeof(int32_t) * (size_t)rings.n);
        char *ring_has_ord = (char *)malloc((size_t)rings.n);
        for (int32_t k = 0; k < rings.n; k++) ring_has_ord[k] = (char)plane_res_ord(s, &rings.p[k], &ring_ord_v[k]);
        for (int32_t k = 0; k < rings.n; k++) {
            const OrcPlane *ring = &rings.p[k];
            if (!ring_has_ord[k]) continue;
            int32_t ring_ord = ring_ord_v[k];
            ResKey rk = {ring->model_serial, ring->chain, ring_ord};
            for (int32_t yi = 0; yi < s->n; yi++) {
                const OrcAtom *y = &s->atoms[yi];
                double dx = y->x - ring->c[0], dy = y->y - ring->c[1], dz = y->z - ring->c[2];
                if (!(dx * dx + dy * dy + dz * dz <= r2)) continue;
                ResKey yk = {y->model_serial, y->chain, y->res_ord};
                if (!should_compare_residues(&cx, &rk, &yk, 0)) continue;
                if (!is_pos_ionizable(y->res_resn, y->name)) continue;
                double q[3]; atom_pos(y, q);
                double dist = plane_point_dist(ring, q), theta = plane_point_angle(ring, q);
                if ((theta <= 30.0) && (dist <= 4.5)) {
                    OrcRow *r = rv_push(&rows);
                    r->model = (uint32_t)ring->model_serial; r->interaction = ORC_CationPi; r->distance = dist;
                    entity_from_ring(&r->from, ring); entity_from_atom(&r->to, y);
                    r->from_atom = -1; r->to_atom = yi;
                }
            }
        }
        /* ring-ring rows (complex.rs:354-405, aromatic.rs:33-64) */
        for (int32_t a = 0; a < rings.n; a++)
            for (int32_t b = 0; b < rings.n; b++) {
                const OrcPlane *k1 = &rings.p[a], *k2 = &rings.p[b];
                if (!(set_has(&cx.lig, k1->chain) && set_has(&cx.rec, k2->chain))) continue;
                if (!ring_has_ord[a] || !ring_has_ord[b]) continue;
                int32_t o1 = ring_ord_v[a], o2 = ring_ord_v[b];
                ResKey r1 = {k1->model_serial, k1->chain, o1}, r2k = {k2->model_serial, k2->chain, o2};
                if (!should_compare_residues(&cx, &r1, &r2k, 1)) continue;
                double v[3] = {k1->c[0] - k2->c[0], k1->c[1] - k2->c[1], k1->c[2] - k2->c[2]};
                double dist = norm3(v);
                int code = -1;
                if (dist <= 6.0) {
                    double theta = plane_point_angle(k1, k2->c), dih = plane_dihedral(k1, k2);
                    if (dih <= 30.0) {
                        if (theta <= 30.0) code = ORC_PiSandwichStacking;
                        else if (theta <= 60.0) code = ORC_PiDisplacedStacking;
                        else if (theta <= 90.0) code = ORC_PiParallelInPlaneStacking;
                    } else if (dih <= 60.0) code = ORC_PiTiltedStacking;
                    else if (dih <= 90.0) {
                        if (theta >= 30.0 && theta < 60.0) code = ORC_PiLStacking;
                        else if (dist <= 5.0) code = ORC_PiTStacking;
                    }
                }
                if (code >= 0) {
                    OrcRow *r = rv_push(&rows);
                    r->model = (uint32_t)k1->model_serial; r->interaction = code; r->distance = dist;
                    entity_from_ring(&r->from, k1); entity_from_ring(&r->to, k2);
                    r->from_atom = -1; r->to_atom = -1;
                }
            }
        free(ring_ord_v); free(ring_has_ord);
        /* side-chain plane statistics (complex.rs:137-174) joined on the residue ids (mod.rs:100-119) */
        for (int64_t k = 0; k < rows.n; k++) {
            OrcRow *r = &rows.r[k];
            const OrcPlane *p1 = find_plane(&scp, r->model, &r->from);
            const OrcPlane *p2 = p1 ? find_plane(&scp, r->model, &r->to) : NULL;
            if (p1 && p2) {
                r->has_sc = 1;
                r->sc_centroid_dist = plane_point_dist(p1, p2->c);
                r->sc_dihedral = plane_dihedral(p1, p2);
                r->sc_centroid_angle = plane_point_angle(p1, p2->c);
            }
        }
        qsort(rows.r, (size_t)rows.n, sizeof(OrcRow), cmp_rows);
    }
    free(pv.p); free(rings.p); free(scp.p); complex_free(&cx);
    if (rc != ORC_OK) { free(rows.r); return rc; }
    *rows_out = rows.r; *n_rows = rows.n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ SAP neighbour sum
 * src/sap.rs:155-204 of the reference: for every side-chain atom x, the sum over the side-chain atoms y (x itself included) with
 * |y - x|^2 <= f64(sap_radius * sap_radius) (rstar locate_within_distance: inclusive; the product is formed in f32, sap.rs:183) of
 * hydrophobicity(resn(y)) * clamp(sasa(y) / max_sc_asa(resn(y)), 0, 1), accumulated in f32 (.sum::<f32>()).  The per-atom SASA itself
 * (src/sasa.rs, arithmetic in the rust-sasa crate) is out of scope: the caller passes it.  Brute force, index order. */
static int sap_tables(const char *resn, float *hyd, float *max_asa) {
    static const struct { const char *n; float h, a; } T[] = {  /* sap.rs:41-64 (Black & Mould minus glycine), sap.rs:77-101 */
        {"ALA", 0.616f - 0.501f, 15.395f}, {"ARG", 0.000f - 0.501f, 124.338f}, {"ASN", 0.236f - 0.501f, 90.303f}, {"ASP", 0.028f - 0.501f, 87.601f},
        {"CYS", 0.680f - 0.501f, 46.456f}, {"GLU", 0.043f - 0.501f, 95.534f}, {"GLN", 0.251f - 0.501f, 99.186f}, {"GLY", 0.000f, 3.229f},
        {"HIS", 0.165f - 0.501f, 96.532f}, {"ILE", 0.943f - 0.501f, 31.448f}, {"LEU", 0.943f - 0.501f, 30.271f}, {"LYS", 0.283f - 0.501f, 61.962f},
        {"MET", 0.738f - 0.501f, 65.233f}, {"PHE", 1.000f - 0.501f, 67.945f}, {"PRO", 0.711f - 0.501f, 17.812f}, {"SER", 0.359f - 0.501f, 39.355f},
        {"THR", 0.450f - 0.501f, 42.648f}, {"TRP", 0.878f - 0.501f, 101.491f}, {"TYR", 0.880f - 0.501f, 94.478f}, {"VAL", 0.825f - 0.501f, 26.702f}};
    char up[8] = {0};
    for (int k = 0; k < 7 && resn[k]; k++) up[k] = (char)toupper((unsigned char)resn[k]);
    for (size_t k = 0; k < sizeof T / sizeof T[0]; k++) if (strcmp(T[k].n, up) == 0) { *hyd = T[k].h; *max_asa = T[k].a; return 1; }
    return 0;
}
float orc_sap_weight(const char *resn, float sasa) {  /* sap.rs:198-209: 0 when the residue has no hydrophobicity value */
    float h, a;
    if (!sap_tables(resn, &h, &a)) return 0.0f;
    float q = sasa / a;
    if (q < 0.0f) q = 0.0f;
    if (q > 1.0f) q = 1.0f;
    return h * q;
}
void orc_sap_neighbor_sum(int64_t n, const double *x, const double *y, const double *z, const uint8_t *sidechain, const float *weight, float sap_radius,
                          float *out) {
    const double r2 = (double)(sap_radius * sap_radius);  /* sap.rs:183 */
    for (int64_t i = 0; i < n; i++) {
        float acc = 0.0f;
        if (sidechain[i])
            for (int64_t j = 0; j < n; j++) {
                if (!sidechain[j]) continue;
                const double dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
                if (dx * dx + dy * dy + dz * dz <= r2) acc += weight[j];
            }
        out[i] = acc;
    }
}

/* ------------------------------------------------------------------ CLI: dump the table as CSV */
#ifdef ORC_MAIN
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.pdb [groups] [vdw_comp] [dist_cutoff]\n", argv[0]); return 2; }
    const char *groups = argc > 2 ? argv[2] : "/";
    double c = argc > 3 ? atof(argv[3]) : 0.1, d = argc > 4 ? atof(argv[4]) : 6.5;
    OrcStructure *s = orc_load_model(argv[1], 0);
    if (!s) { fprintf(stderr, "load failed: %s\n", orc_last_error()); return 1; }
    OrcRow *rows; int64_t n;
    int rc = orc_get_contacts(s, groups, c, d, &rows, &n);
    if (rc != ORC_OK) { fprintf(stderr, "error %d: %s\n", rc, orc_last_error()); return 1; }
    printf("model,interaction,distance,from_chain,from_resn,from_resi,from_insertion,from_altloc,from_atomn,from_atomi,"
           "to_chain,to_resn,to_resi,to_insertion,to_altloc,to_atomn,to_atomi,sc_centroid_dist,sc_dihedral,sc_centroid_angle\n");
    for (int64_t k = 0; k < n; k++) {
        const OrcRow *r = &rows[k];
        printf("%u,%s,%.9g,%s,%s,%d,%s,%s,%s,%d,%s,%s,%d,%s,%s,%s,%d,", r->model, orc_interaction_name(r->interaction),
               (double)(float)r->distance, r->from.chain, r->from.resn, r->from.resi, r->from.insertion, r->from.altloc, r->from.atomn,
               r->from.atomi, r->to.chain, r->to.resn, r->to.resi, r->to.insertion, r->to.altloc, r->to.atomn, r->to.atomi);
        if (r->has_sc) printf("%.9g,%.9g,%.9g\n", (double)(float)r->sc_centroid_dist, (double)(float)r->sc_dihedral, (double)(float)r->sc_centroid_angle);
        else printf(",,\n");
    }
    free(rows); orc_free_structure(s);
    return 0;
}
#endif
