// pt_wavefront.hip -- placeholder until the wavefront pipeline lands (next commit).
#include "pt_wavefront.h"
#include "pt_device.h"
namespace hrt {
bool wavefront_supports(const SceneView&, const HrptPathTracerConstants&) { return false; }
hipError_t wavefront_render(WavefrontState&, const SceneView&, const HrptPathTracerConstants&, uint32_t, float4*, float4*, uint32_t, uint32_t,
                            TileRect, DeviceCounters*, hipStream_t, std::string& error) { error = "not built"; return hipErrorNotSupported; }
void wavefront_release(WavefrontState&) {}
void wavefront_trace_timing(const WavefrontState& st, float* ms, uint32_t* n) { *ms = st.traceMs; *n = st.traceLaunches; }
}
