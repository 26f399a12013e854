/* Minimal C (not C++) caller of the boundary library: proves include/s360.h is plain C and that the host-only
 * entry points work without a GPU.  Build + run (tests/test_abi.py does exactly this):
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_abi_layout.c -Lsplatter360_amd -ls360 \
 *       -Wl,-rpath,$PWD/splatter360_amd -o /tmp/c_abi_layout && /tmp/c_abi_layout
 * A real caller continues with hipMalloc'd buffers and s360_forward / s360_backward. */
#include <stdio.h>
#include <string.h>

#include "s360.h"

int main(void) {
    S360Params prm;
    S360Layout lay;
    memset(&prm, 0, sizeof prm);
    prm.P = 1048576;
    prm.V = 6;
    prm.H = prm.W = 256;
    prm.sh_degree = 4;
    prm.M = 25;
    prm.flags = S360_FLAG_SHARED_CAMPOS | S360_FLAG_COV9 | S360_FLAG_SH_CHANNEL_MAJOR;
    prm.max_instances = 9699328u; /* 1.5 * P * V + 256 Ki */
    if (s360_abi_version() != S360_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch: header %d, library %d\n", S360_ABI_VERSION, s360_abi_version());
        return 2;
    }
    int rc = s360_layout(&prm, &lay);
    if (rc != 0) {
        fprintf(stderr, "s360_layout: %s\n", s360_error_string(rc));
        return 1;
    }
    printf("abi %d forward_workspace_bytes %zu backward_workspace_bytes %zu keys_offset %zu\n", s360_abi_version(),
           lay.total_bytes, lay.backward_bytes, lay.keys);
    prm.V = 99; /* invalid: must be rejected with an error code, not a crash */
    rc = s360_layout(&prm, &lay);
    printf("bad V -> %d (%s)\n", rc, s360_error_string(rc));
    return rc == 0 ? 3 : 0;
}
