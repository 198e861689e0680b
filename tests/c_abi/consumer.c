/* A plain C11 consumer of include/arpeggia_amd.h -- what a C (or, through `extern "C"` + #[repr(C)], a Rust) host of the reference
 * would compile against: the seam is `pub fn get_contacts(&PDB, &str, f64, f64) -> DataFrame` (src/contacts/mod.rs:61, re-exported
 * src/lib.rs:28; PyO3 wrapper src/python.rs:31-56).
 *
 *   consumer --abi            prints the layout of every struct of the header as JSON (no device call): tests/test_host_cpu.py compares it
 *                             with the ctypes mirror in arpeggia_amd/_lib.py, field by field
 *   consumer FILE.pdb         INTEGRATION.md section 3 verbatim: load the model -> arp_get_contacts -> columns -> Arrow C Data export ->
 *                             release; prints "<rows> rows" (532 for test-data/1ubq.pdb, python/tests/test_arpeggia.py:35)
 *
 * Built by tests/conftest.py with `gcc -std=c11 -Wall -Werror -Iinclude` and linked against libarpeggia_amd.so.
 */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "arpeggia_amd.h"

/* The layout a binder hard-codes (the #[repr(C)] structs).  A reordered or resized field breaks the build here. */
#define AT(T, F, OFF) _Static_assert(offsetof(T, F) == (OFF), #T "." #F " moved")
AT(arp_atoms, n, 0); AT(arp_atoms, x, 8); AT(arp_atoms, y, 16); AT(arp_atoms, z, 24); AT(arp_atoms, attr, 32); AT(arp_atoms, res_ord, 40);
AT(arp_atoms, chain_rank, 48); AT(arp_atoms, model, 56); AT(arp_atoms, res_id, 64); AT(arp_atoms, n_res, 72); AT(arp_atoms, res_h_ptr, 80);
AT(arp_atoms, res_h_idx, 88); AT(arp_atoms, res_cb, 96); AT(arp_atoms, res_sg, 104); AT(arp_atoms, location, 112); AT(arp_atoms, reserved, 116);
_Static_assert(sizeof(arp_atoms) == 120, "arp_atoms size");
AT(arp_params, vdw_comp, 0); AT(arp_params, dist_cutoff, 8); AT(arp_params, cov_radius, 16); AT(arp_params, vdw_radius, 144);
AT(arp_params, h_vdw_radius, 272); AT(arp_params, flags, 280); AT(arp_params, reserved, 284);
_Static_assert(sizeof(arp_params) == 288, "arp_params size");
AT(arp_pair, i, 0); AT(arp_pair, j, 4); AT(arp_pair, dist, 8); AT(arp_pair, kind, 12);
_Static_assert(sizeof(arp_pair) == 16, "arp_pair size");
AT(arp_pairs, n, 0); AT(arp_pairs, data, 8); AT(arp_pairs, location, 16); AT(arp_pairs, reserved, 20);
_Static_assert(sizeof(arp_pairs) == 24, "arp_pairs size");
AT(arp_records, n, 0); AT(arp_records, x, 8); AT(arp_records, y, 16); AT(arp_records, z, 24); AT(arp_records, occupancy, 32); AT(arp_records, serial, 40);
AT(arp_records, resi, 48); AT(arp_records, model_serial, 56); AT(arp_records, name, 64); AT(arp_records, resn, 72); AT(arp_records, chain, 80);
AT(arp_records, altloc, 88); AT(arp_records, icode, 96); AT(arp_records, element, 104); AT(arp_records, res_ord, 112); AT(arp_records, res_id, 120);
_Static_assert(sizeof(arp_records) == 128, "arp_records size");
/* Arrow C Data Interface (ABI-stable by specification) */
AT(struct ArrowSchema, format, 0); AT(struct ArrowSchema, name, 8); AT(struct ArrowSchema, metadata, 16); AT(struct ArrowSchema, flags, 24);
AT(struct ArrowSchema, n_children, 32); AT(struct ArrowSchema, children, 40); AT(struct ArrowSchema, dictionary, 48); AT(struct ArrowSchema, release, 56);
AT(struct ArrowSchema, private_data, 64);
_Static_assert(sizeof(struct ArrowSchema) == 72, "ArrowSchema size");
AT(struct ArrowArray, length, 0); AT(struct ArrowArray, null_count, 8); AT(struct ArrowArray, offset, 16); AT(struct ArrowArray, n_buffers, 24);
AT(struct ArrowArray, n_children, 32); AT(struct ArrowArray, buffers, 40); AT(struct ArrowArray, children, 48); AT(struct ArrowArray, dictionary, 56);
AT(struct ArrowArray, release, 64); AT(struct ArrowArray, private_data, 72);
_Static_assert(sizeof(struct ArrowArray) == 80, "ArrowArray size");
_Static_assert(ARP_N_INTERACTIONS == 19 && ARP_HydrophobicContact == 18 && ARP_StericClash == 0, "interaction codes follow structs.rs:6-51");

#define FIELD(T, F) printf("%s\"%s\": [%zu, %zu]", first ? "" : ", ", #F, offsetof(T, F), sizeof(((T *)0)->F)), first = 0
#define OPEN(T) printf("%s\"%s\": {\"sizeof\": %zu, \"fields\": {", any ? ", " : "", #T, sizeof(T)), any = 1, first = 1
#define CLOSE() printf("}}")

static int print_abi(void) {
    int any = 0, first = 1;
    printf("{\"api_version\": %d, ", (int)arp_api_version());  /* (proves the program is linked against the library) */
    printf("\"structs\": {");
    OPEN(arp_atoms);
    FIELD(arp_atoms, n); FIELD(arp_atoms, x); FIELD(arp_atoms, y); FIELD(arp_atoms, z); FIELD(arp_atoms, attr); FIELD(arp_atoms, res_ord);
    FIELD(arp_atoms, chain_rank); FIELD(arp_atoms, model); FIELD(arp_atoms, res_id); FIELD(arp_atoms, n_res); FIELD(arp_atoms, res_h_ptr);
    FIELD(arp_atoms, res_h_idx); FIELD(arp_atoms, res_cb); FIELD(arp_atoms, res_sg); FIELD(arp_atoms, location); FIELD(arp_atoms, reserved);
    CLOSE();
    OPEN(arp_params);
    FIELD(arp_params, vdw_comp); FIELD(arp_params, dist_cutoff); FIELD(arp_params, cov_radius); FIELD(arp_params, vdw_radius); FIELD(arp_params, h_vdw_radius);
    FIELD(arp_params, flags); FIELD(arp_params, reserved);
    CLOSE();
    OPEN(arp_pair);
    FIELD(arp_pair, i); FIELD(arp_pair, j); FIELD(arp_pair, dist); FIELD(arp_pair, kind);
    CLOSE();
    OPEN(arp_pairs);
    FIELD(arp_pairs, n); FIELD(arp_pairs, data); FIELD(arp_pairs, location); FIELD(arp_pairs, reserved);
    CLOSE();
    OPEN(arp_records);
    FIELD(arp_records, n); FIELD(arp_records, x); FIELD(arp_records, y); FIELD(arp_records, z); FIELD(arp_records, occupancy); FIELD(arp_records, serial);
    FIELD(arp_records, resi); FIELD(arp_records, model_serial); FIELD(arp_records, name); FIELD(arp_records, resn); FIELD(arp_records, chain);
    FIELD(arp_records, altloc); FIELD(arp_records, icode); FIELD(arp_records, element); FIELD(arp_records, res_ord); FIELD(arp_records, res_id);
    CLOSE();
    printf("}}\n");
    return 0;
}

#define TRY(call)                                                                                   \
    do {                                                                                            \
        arp_status st_ = (call);                                                                    \
        if (st_ != ARP_OK) { fprintf(stderr, "%s -> %s: %s\n", #call, arp_strerror(st_), arp_last_error()); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: consumer --abi | consumer FILE.pdb\n"); return 2; }
    if (strcmp(argv[1], "--abi") == 0) return print_abi();

    /* INTEGRATION.md section 3 */
    arp_structure *s = NULL; arp_table *t = NULL; arp_context *ctx = NULL;
    TRY(arp_context_create(0, &ctx));
    TRY(arp_structure_load(argv[1], /*ignore_zero_occupancy=*/0, &s));   /* utils.rs:51-63 load_model */
    TRY(arp_get_contacts(ctx, s, "/", 0.1, 6.5, &t));                    /* mod.rs:61 */
    const uint64_t rows = arp_table_rows(t);
    int32_t w = 0;
    const float *dist = (const float *)arp_table_column(t, "distance", &w);  /* 20 reference columns + sc_valid, from_atom, to_atom */
    const int32_t *code = (const int32_t *)arp_table_column(t, "interaction", &w);
    if (!dist || !code) { fprintf(stderr, "missing column\n"); return 1; }
    double dmax = 0.0;
    for (uint64_t r = 0; r < rows; r++) {
        if (!(dist[r] >= 0.0f) || code[r] < 0 || code[r] >= ARP_N_INTERACTIONS) { fprintf(stderr, "bad row %llu\n", (unsigned long long)r); return 1; }
        if (dist[r] > dmax) dmax = dist[r];
    }
    /* the same table through the Arrow C Data Interface: what pyo3-polars hands to Python in the reference (python.rs:55) */
    struct ArrowArray arr; struct ArrowSchema sch;
    memset(&arr, 0, sizeof arr); memset(&sch, 0, sizeof sch);
    TRY(arp_table_export_arrow(t, &arr, &sch));
    if ((uint64_t)arr.length != rows || arr.n_children != 20 || sch.n_children != 20 || strcmp(sch.format, "+s") != 0) {
        fprintf(stderr, "arrow export: length %lld, %lld columns, format %s\n", (long long)arr.length, (long long)arr.n_children, sch.format);
        return 1;
    }
    printf("columns:");
    for (int64_t k = 0; k < sch.n_children; k++) printf(" %s", sch.children[k]->name);
    printf("\n");
    arp_table_free(t);              /* the exported arrays own copies: the table may go first */
    arr.release(&arr); sch.release(&sch);
    arp_structure_free(s);
    arp_context_destroy(ctx);
    printf("%llu rows, max distance %.3f, first interaction %s\n", (unsigned long long)rows, dmax, rows ? arp_interaction_name(0) : "-");
    return 0;
}
