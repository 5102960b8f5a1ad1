"""INTEGRATION.md's C / C++ snippets must compile against include/crx.h: every call they show has the argument list the header
declares (a signature that drifts makes the document lie).  Each ```c / ```cpp block becomes the body of a function behind a prelude
that declares the names the snippets leave to the reader; g++ -fsyntax-only checks it."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
#include <vector>
#include "crx.h"
extern int n, T, ncourse, nx, nob, SIM_LOOP;
extern float *xEst, *PEst, *ud, *hxEst, *Q, *R, *cx_dev, *cy_dev, *cyaw_dev, *ck_dev, *sp_dev, *state_dev, *traj_hist_dev,
             *coef_dev, *ob_dev, *hist_dev;
extern int *ticks_done_dev, *target_ind_dev, *ticks_dev, *status_dev;
extern float goal_x, goal_y, target_speed;
extern void* stream;
extern std::vector<float> wx, wy, r_x_, sp, ryaw, rcurvature;
// what the pipelined-host snippet (section 5b) leaves to the reader: HIP's stream / event calls and the host's own buffers
typedef void* hipStream_t; typedef void* hipEvent_t;
int hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned); int hipEventRecord(hipEvent_t, hipStream_t);
const int DEPTH = 6;
extern int m, rounds; extern crx_course course;
extern float *est[DEPTH], *err[DEPTH], *xref[DEPTH], *sol[DEPTH]; extern int *tind[DEPTH], *status[DEPTH]; extern double* cost[DEPTH];
void gather_every_eighth(float*, const float*, hipStream_t);
// ... and the C round / multi-GPU snippets (section 5c)
#include <cstdlib>
extern crx_course course_dev; extern float *x0_dev, *P0_dev, **z_dev, **ud_dev, *hxEst_dev, *xEst_dev, *all_xEst_dev; extern int rank, world;
void consume(const float*, const int*, int, void*);
'''


def snippets():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```(?:c|cpp)\n(.*?)```", text, flags=re.S)


def test_integration_md_snippets_compile(tmp_path):
    blocks = snippets()
    assert len(blocks) >= 3
    src = PRELUDE
    for i, b in enumerate(blocks):
        body = "\n".join(l for l in b.splitlines() if l.strip() != "...")
        # names a snippet defines itself must not clash with the prelude: each block is its own function
        decl = ""
        if "r_x.data()" in body and "std::vector<float> r_x" not in body:
            decl += "std::vector<float> r_x, r_y;\n"
        if re.search(r"\bz\b", body) and "float* z" not in body:
            decl += "float* z = nullptr;\n"
        src += f"\nvoid snippet_{i}() {{\n{decl}{body}\n}}\n"
    f = tmp_path / "snippets.cpp"
    f.write_text(src)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
                        "-I", os.path.join(ROOT, "include"), str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:3000]
