/* direct_image.S -- the step kernels as a bare gfx950 code object (step_kernels.hip compiled --offload-device-only by build.py),
 * linked into libenv2048.so as read-only data: direct_dispatch.hpp loads it through ROCr for the queues it drives itself.
 * ENV2048_DIRECT_IMAGE = path of the .hsaco (set by build.py). */
    .section .rodata
    .globl env2048_direct_image
    .globl env2048_direct_image_end
    .hidden env2048_direct_image
    .hidden env2048_direct_image_end
    .balign 4096
env2048_direct_image:
    .incbin ENV2048_DIRECT_IMAGE
env2048_direct_image_end:
    .byte 0
    .section .note.GNU-stack,"",@progbits
