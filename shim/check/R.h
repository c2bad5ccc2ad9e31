/* see Rinternals.h in this directory: syntax-check declarations only */
#include <string.h>
#include <stdint.h>
#include <math.h>
