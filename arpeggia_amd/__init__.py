"""arpeggia_amd: MI355X-native drop-in for the `contacts` path of y1zhou/arpeggia.

Public surface mirrors the reference (src/lib.rs:20-34, src/python.rs:31-56) for this one path:
contacts(), get_contacts(), load_model(), parse_groups().  Importing this package loads libarpeggia_amd.so and
fails loudly if the HIP extension has not been built -- there is no CPU fallback.
"""
from .api import (  # noqa: F401
    ArpeggiaError, Context, Structure, PAIR_DTYPE, TABLE_COLUMNS, atomic_contacts_batch, atoms_from_arrays, contacts, contacts_batch, debug_set, default_params,
    device_count, get_contacts, load_model, parse_groups, sap_neighbor_sum, sap_weight,
)
from ._lib import ATTR, INTERACTIONS  # noqa: F401

__version__ = "0.1.0"
