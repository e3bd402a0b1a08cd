/* TEST INFRASTRUCTURE (oracle): stages whose restatement is not written yet. */
#include "ostate.h"
#include <stdio.h>
#include <stdlib.h>
#define TODO(name) { fprintf(stderr, "oracle/c: stage %s not restated yet\n", name); abort(); }
void orc_diapfl(OState *S, int n, int nn, int k1n) TODO("diapfl")

