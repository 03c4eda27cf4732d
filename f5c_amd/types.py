"""POD mirrors of the f5c data contract (reference src/f5c.h) as numpy dtypes."""
import numpy as np

# event_t f5c.h:129-136 (24 B with tail padding)
EVENT_DT = np.dtype([("start", "<u8"), ("length", "<f4"), ("mean", "<f4"), ("stdv", "<f4")], align=True)
# model_t f5c.h:147-155 (CACHED_LOG)
MODEL_DT = np.dtype([("level_mean", "<f4"), ("level_stdv", "<f4"), ("level_log_stdv", "<f4")])
# AlignedPair f5c.h:181-184
PAIR_DT = np.dtype([("ref_pos", "<i4"), ("read_pos", "<i4")])
# scalings_t f5c.h:158-172 (CACHED_LOG)
SCAL_DT = np.dtype([("scale", "<f4"), ("shift", "<f4"), ("var", "<f4"), ("log_var", "<f4")])
# abea_read_diag include/abea.h
DIAG_DT = np.dtype([("sum_emission", "<f8"), ("n_aligned", "<i4"), ("best_event", "<i4"),
                    ("max_score", "<f4"), ("max_gap", "<i4"), ("spanned", "<i4"), ("flags", "<i4"),
                    ("pad", "<i4")], align=True)
assert EVENT_DT.itemsize == 24 and MODEL_DT.itemsize == 12 and PAIR_DT.itemsize == 8
assert SCAL_DT.itemsize == 16 and DIAG_DT.itemsize == 40
