/* queue_churn.c -> tools/libqueue_churn.so: create and destroy a user-mode queue on the first GPU agent `count` times, `pause_us`
 * apart -- called from a background thread of tools/fuzz_direct.py (--churn) to move the XCD placement of the process's other
 * queues WHILE direct-route sequences run (the library must notice and re-run them: env2048_direct_stats.reruns).
 * gcc -O2 -shared -fPIC tools/queue_churn.c -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o tools/libqueue_churn.so */
#include <hsa/hsa.h>
#include <stdint.h>
#include <unistd.h>

static hsa_agent_t g_gpu;
static int g_have;

static hsa_status_t visit(hsa_agent_t agent, void *data)
{
    hsa_device_type_t type;
    (void)data;
    if (!g_have && hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_GPU) {
        g_gpu = agent;
        g_have = 1;
    }
    return HSA_STATUS_SUCCESS;
}

int queue_churn(int count, int pause_us)
{
    int made = 0;
    if (hsa_init() != HSA_STATUS_SUCCESS) return -1;
    hsa_iterate_agents(visit, 0);
    if (!g_have) return -2;
    for (int i = 0; i < count; ++i) {
        hsa_queue_t *q = 0;
        if (hsa_queue_create(g_gpu, 64, HSA_QUEUE_TYPE_SINGLE, 0, 0, UINT32_MAX, UINT32_MAX, &q) == HSA_STATUS_SUCCESS) {
            ++made;
            usleep(pause_us);
            hsa_queue_destroy(q);
        }
        usleep(pause_us);
    }
    return made;
}
