/*
 * Determinism shim for the reference build ONLY (oracle/_ref/libref_cpu.so).
 *
 * cpu::calc_optical_flow / gpu::calc_opt_flow read bytes of a malloc'd buffer
 * that was never written (OptFlowCPU.cpp:320 + :247/:270-273; SURVEY.md 8c,
 * determinism caveat 2).  Linking the reference objects with
 * -Wl,--wrap=malloc routes their malloc calls here, so those bytes are zero on
 * every run.  This adds no header, library or generated code the reference
 * needs in order to build; it only pins what its uninitialised reads return.
 */
#include <stdlib.h>
void *__wrap_malloc(size_t n) { return calloc(1, n ? n : 1); }
