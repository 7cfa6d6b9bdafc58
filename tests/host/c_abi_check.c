/* c_abi_check.c — include/lio_c.h from a PLAIN C translation unit (gcc -std=c99 -pedantic-errors), linked against the
 * library under test.  Proves the boundary is C-clean: the header needs no C++ feature, and every declared entry point
 * resolves with C linkage.  tests/test_abi.py generates lio_symbols.inc (one LIO_SYM(name) line per declaration in the
 * header), compiles this file and runs it.  No data-path call is made: it runs on a box without a GPU. */
#include "lio_c.h"

#include <stdio.h>
#include <string.h>

typedef void (*lio_any_fn)(void);
struct sym { const char *name; lio_any_fn fn; };

#define LIO_SYM(x) {#x, (lio_any_fn)x},
static const struct sym table[] = {
#include "lio_symbols.inc"
    {0, 0}};

int main(int argc, char **argv) {
  size_t n = 0;
  lio_pp_config pp;
  lio_est_config ec;
  const char *want = argc > 1 ? argv[1] : 0;
  while (table[n].name) {
    if (!table[n].fn) { fprintf(stderr, "null symbol %s\n", table[n].name); return 2; }
    ++n;
  }
  if (want && strcmp(lio_backend(), want) != 0) { fprintf(stderr, "backend %s != %s\n", lio_backend(), want); return 3; }
  lio_pp_default_config(&pp);
  if (pp.num_scan_subregions != 8 || pp.max_corner_sharp != 2) return 4;
  lio_est_default_config(&ec);
  if (ec.window_size != 15 || ec.opt_window_size != 5) return 5;
  /* plain-pointer, size-only signatures: a malformed message is refused without touching a device */
  {
    float msg[16];
    size_t nc = 0, ns = 0, nf = 0;
    memset(msg, 0, sizeof msg);
    msg[8] = -1.0f;
    if (lio_compact_decode(msg, 4, 0, &nc, &ns, &nf) != LIO_ERR_ARG) return 6;
  }
  printf("%s: %lu symbols\n", lio_backend(), (unsigned long)n);
  return 0;
}
