/* The boundary from plain C (tests/test_host_logic.py compiles this with gcc -std=c99 -pedantic and runs it without a GPU):
 * the header is valid C, the library links with -ldig_hip, the query entry points answer and a launch entry point rejects
 * a null argument block through its return code and dig_last_error(). */
#include <stdio.h>
#include "dig_hip.h"
int main(void) {
    int bn = 0, kc = 0;
    if (dig_version() < 100) return 1;
    if (dig_conv3x3_config(512, 0, &bn, &kc) != 0 || bn != 128 || kc != 16) return 2;
    dig_sched_axpby_args a = {0};
    if (dig_sched_axpby(&a, 0) == 0) return 3;
    printf("%d %d %s\n", bn, kc, dig_last_error());
    return 0;
}
