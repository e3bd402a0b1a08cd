/* TEST INFRASTRUCTURE (oracle): stages whose restatement is not written yet. */
#include "ostate.h"
#include <stdio.h>
#include <stdlib.h>
#define TODO(name) { fprintf(stderr, "oracle/c: stage %s not restated yet\n", name); abort(); }

