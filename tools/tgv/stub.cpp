#include <stdio.h>
extern "C" int yy_tower_set_err(int c, const char *m) { fprintf(stderr, "tower error %d: %s\n", c, m); return c; }
