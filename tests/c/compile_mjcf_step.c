/* A C host of the boundary, nothing but include/rgstep.h: an MJCF document (file argv[2]) goes through rg_compile_mjcf -- what
 * MujocoXML.build -> mujoco_py.load_model_from_xml(xml_string) is to the reference (/root/reference/robogym/mujoco/mujoco_xml.py:249-260) --, a batch of
 * two envs is created from the model, stepped argv[3] times with zero actions (rg_batch_step_ex) and the first env's qpos is printed.
 * TEST PROGRAM (tests/test_compile_mjcf.py builds it with gcc and runs it against the library given as argv[1], dlopen'ed: the product library on a GPU box,
 * the emulation-harness build of the same source on CPU).  Exit status 0 = every call succeeded. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rgstep.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <librgstep.so> <model.xml> <nsteps> [bad]\n", argv[0]); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(rg_compile_mjcf) SYM(rg_model_dims) SYM(rg_batch_create) SYM(rg_batch_copy) SYM(rg_batch_step_ex) SYM(rg_sync) SYM(rg_batch_free) SYM(rg_model_free) SYM(rg_last_error)
  FILE* f = fopen(argv[2], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char* xml = (char*)malloc(n + 1);
  if (fread(xml, 1, n, f) != (size_t)n) return 2;
  xml[n] = 0; fclose(f);
  char err[600] = "";
  rg_model* m = p_rg_compile_mjcf(xml, NULL, err, sizeof err);
  if (!m) { printf("compile failed: %s\n", err); return argc > 4 ? 0 : 1; }   /* ("bad": the caller expects the compiler's message) */
  int dims[5];
  if (p_rg_model_dims(m, dims) != 0) return 1;
  printf("dims nq %d nv %d nu %d nbody %d nsite %d\n", dims[0], dims[1], dims[2], dims[3], dims[4]);
  rg_batch* b = p_rg_batch_create(m, 2, 0);
  if (!b) { printf("rg_batch_create: %s\n", p_rg_last_error()); return 1; }
  rg_step_args a; memset(&a, 0, sizeof a);
  a.nsubsteps = 10; a.nforward_ticks = 3;
  const int nsteps = atoi(argv[3]);
  for (int k = 0; k < nsteps; k++) if (p_rg_batch_step_ex(b, &a) != 0) { printf("rg_batch_step_ex: %s\n", p_rg_last_error()); return 1; }
  if (p_rg_sync(NULL) != 0) return 1;
  float* q = (float*)malloc(sizeof(float) * 2 * dims[0]);
  if (p_rg_batch_copy(b, RG_F_QPOS, q, /*to_batch*/ 0, /*ptr_is_device*/ 0) != 0) { printf("rg_batch_copy: %s\n", p_rg_last_error()); return 1; }
  printf("qpos");
  for (int i = 0; i < dims[0]; i++) printf(" %.9g", q[i]);
  printf("\n");
  p_rg_batch_free(b); p_rg_model_free(m);
  return 0;
}
