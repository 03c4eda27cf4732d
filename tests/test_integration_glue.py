"""The f5c-side glue (integration/abea_glue.c, quoted in INTEGRATION.md), compile-checked against the REAL reference headers (src/f5c.h,
src/f5cmisc.h) — build container only: /root/reference is absent on the GPU box.  htslib is not in the image, so
opaque typedef stubs for the six htslib types f5c.h names are generated into a temp dir; nothing of this is committed
or shipped, and nothing is linked or run (-fsyntax-only).  Also asserts the POD layouts the shim's casts rely on."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "f5c.h")),
                                reason="reference headers not mounted (GPU box)")

LAYOUT = r'''
#include <cstddef>
#include "f5c.h"
#include "f5cmisc.h"
#include "abea_f5c_shim.h"
// the casts in the glue: (abea_f5c_event_table*)db->et, (abea_scalings_t*)db->scalings, (abea_pair_t**)..., (const abea_model_t*)core->model
static_assert(sizeof(event_table) == sizeof(abea_f5c_event_table), "event_table size (f5c.h:139-144)");
static_assert(offsetof(event_table, n) == offsetof(abea_f5c_event_table, n), "event_table.n");
static_assert(offsetof(event_table, start) == offsetof(abea_f5c_event_table, start), "event_table.start");
static_assert(offsetof(event_table, end) == offsetof(abea_f5c_event_table, end), "event_table.end");
static_assert(offsetof(event_table, event) == offsetof(abea_f5c_event_table, event), "event_table.event");
static_assert(sizeof(event_t) == sizeof(abea_event_t) && offsetof(event_t, mean) == offsetof(abea_event_t, mean), "event_t (f5c.h:129-136)");
static_assert(offsetof(event_t, start) == offsetof(abea_event_t, start) && offsetof(event_t, length) == offsetof(abea_event_t, length), "event_t");
static_assert(sizeof(model_t) == sizeof(abea_model_t) && offsetof(model_t, level_stdv) == offsetof(abea_model_t, level_stdv), "model_t (f5c.h:147-155)");
static_assert(offsetof(model_t, level_log_stdv) == offsetof(abea_model_t, level_log_stdv), "model_t needs CACHED_LOG");
static_assert(sizeof(scalings_t) == sizeof(abea_scalings_t) && offsetof(scalings_t, shift) == offsetof(abea_scalings_t, shift), "scalings_t (f5c.h:158-172)");
static_assert(offsetof(scalings_t, var) == offsetof(abea_scalings_t, var) && offsetof(scalings_t, log_var) == offsetof(abea_scalings_t, log_var), "scalings_t");
static_assert(sizeof(AlignedPair) == sizeof(abea_pair_t) && offsetof(AlignedPair, read_pos) == offsetof(abea_pair_t, read_pos), "AlignedPair (f5c.h:181-184)");
static_assert(sizeof(index_pair_t) == sizeof(abea_index_pair_t) && offsetof(index_pair_t, stop) == offsetof(abea_index_pair_t, stop), "index_pair_t (f5c.h:187-190)");
static_assert(ALN_BANDWIDTH == ABEA_BANDWIDTH && MAX_KMER_SIZE == ABEA_MAX_KMER_SIZE, "compile-time constants (f5c.h:30,34)");
static_assert(FAILED_CALIBRATION == ABEA_FAILED_CALIBRATION && FAILED_ALIGNMENT == ABEA_FAILED_ALIGNMENT &&
              FAILED_QUALITY_CHK == ABEA_FAILED_QUALITY_CHK, "read_stat_flag bits (f5c.h:66-68)");
// the prototypes the glue defines
void (*p_init)(core_t*) = init_cuda;
void (*p_free)(core_t*) = free_cuda;
void (*p_align)(core_t*, db_t*) = align_cuda;
'''


def test_glue_compiles_against_the_reference_headers(tmp_path):
    stub = tmp_path / "stub" / "htslib"
    stub.mkdir(parents=True)
    (stub / "hts.h").write_text("#pragma once\ntypedef struct htsFile htsFile; typedef struct hts_idx_t hts_idx_t; "
                                "typedef struct hts_itr_t hts_itr_t;\n")
    (stub / "sam.h").write_text("#pragma once\ntypedef struct htsFile samFile; typedef struct bam1_t bam1_t; "
                                "typedef struct bam_hdr_t bam_hdr_t; typedef struct sam_hdr_t sam_hdr_t;\n")
    (stub / "faidx.h").write_text("#pragma once\ntypedef struct faidx_t faidx_t;\n")
    # the glue is a FILE a maintainer copies (integration/abea_glue.c); INTEGRATION.md quotes its two parts verbatim
    glue = open(os.path.join(ROOT, "integration", "abea_glue.c")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```c\n(// src/abea_glue\.c — .*?)```", md, re.S)
    assert m, "INTEGRATION.md lost its glue block"
    m2 = re.search(r"```c\n(// src/abea_glue\.c, continued.*?)```", md, re.S)      # the process_db_rsq chain (round 4)
    assert m2, "INTEGRATION.md lost its process_db_rsq glue block"
    assert m.group(1) in glue and m2.group(1) in glue, "INTEGRATION.md no longer quotes integration/abea_glue.c"
    (tmp_path / "abea_glue.c").write_text(glue)
    (tmp_path / "layout.cpp").write_text(LAYOUT)
    inc = ["-I", str(tmp_path / "stub"), "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "slow5lib", "include"),
           "-I", os.path.join(ROOT, "include")]
    # the reference compiles its .c files as C++11 with HAVE_CUDA on the GPU build (Makefile:5-8, 40-46)
    base = ["g++", "-x", "c++", "-std=c++11", "-fsyntax-only", "-DHAVE_CUDA=1", "-Wall", "-Werror=return-type"]
    for src in ("abea_glue.c", "layout.cpp"):
        r = subprocess.run(base + inc + [str(tmp_path / src)], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n{r.stderr[-3000:]}"


def test_makefile_patch_applies_and_builds_the_glue(tmp_path):
    """integration/f5c_makefile.patch against the reference's Makefile: applies cleanly to a copy, and `make -n abea=1` compiles
    src/abea_glue.c with the reference's own rule, defines HAVE_CUDA, links -labea_hip and builds none of the .cu files."""
    import shutil
    shutil.copy(os.path.join(REF, "Makefile"), tmp_path / "Makefile")
    r = subprocess.run(["patch", "-p1", "-i", os.path.join(ROOT, "integration", "f5c_makefile.patch")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    (tmp_path / "src").mkdir()
    shutil.copy(os.path.join(ROOT, "integration", "abea_glue.c"), tmp_path / "src" / "abea_glue.c")     # what the maintainer does
    for h in ("f5c.h", "f5cmisc.h"):
        (tmp_path / "src" / h).write_text("")                                # prerequisites of the rule; -n runs nothing
    r = subprocess.run(["make", "-n", "-B", "abea=1", f"ABEA_ROOT={ROOT}", "build/abea_glue.o"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if "abea_glue.c" in ln][0]
    assert "-DHAVE_CUDA=1" in line and f"-I {ROOT}/include" in line and "-x c++" in line
    mk = (tmp_path / "Makefile").read_text()
    assert "-labea_hip" in mk and "$(BUILD_DIR)/abea_glue.o" in mk
    r = subprocess.run(["make", "-n", "-p", "abea=1", f"ABEA_ROOT={ROOT}"], cwd=tmp_path, capture_output=True, text=True)
    objs = [ln for ln in r.stdout.splitlines() if ln.startswith("OBJ ")][0]
    assert "abea_glue.o" in objs and "f5c_cuda.o" not in objs and "gpurocmcode" not in objs and "align_cuda.o" not in objs
