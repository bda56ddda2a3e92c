/* wf_types.h -- host/device-shared descriptors of the wavefront pipeline's queues (see wf_kernels.inc). */
#pragma once
#include <hip/hip_runtime.h>
#include "pt_scene.h"

struct WfQueue {      /* columns of one queue; unused columns may be null for a feature set */
    float4 *a, *b, *c, *d;             /* a = (ro, seed)  b = (rd, weight)  c = (T, bits sampleId)  d = (L, bits flags|bounce<<16; flag bit 15 = WF_FLAG_SSS) */
    float4 *e, *f, *g, *h, *i, *j, *k, *l; /* shadow queue: a = (nextOrigin, seed) b = (prevDir, weight); e = (shadow origin, nDotD)
                                          f = (shadow dir, roughness) g = (N, specCoef) h = (Nd, bits matFlags) j = (albedo, 0) k = (contrib, 0)
                                          FEAT_SSS: i = (absorption, curIOR) l = (scattering, newIOR).
                                          extension queues with FEAT_SSS, entries in the interior walk (WF_FLAG_SSS): e = (absorption, maxTravel)
                                          f = (curIOR, newIOR, roughness, bits sssStep) */
    uint32_t* segCount;                /* [numSegments]: live entries of each segment */
};
#define WF_FLAG_SSS 0x8000u            /* the entry's pending ray is a step of the SSS interior walk (ST_SSS), not a bounce ray */
#define WF_SORT_KEYS 64u               /* material-sorted shading: 0 = miss, 1 = interior-walk step, 2 + materialIndex % 62 */
struct WfHits { float4* tuv_prim; uint32_t* geom; }; /* (t, u, v, bits prim); t = MAX_T on a miss */

/* A queue is cut into numSegments segments of segCapacity slots.  One workgroup processes one segment at a time and
 * appends the survivors to the SAME segment of the output queue (paths never multiply, so it cannot overflow),
 * compacting with wave ballots + one LDS atomic per wave: no global atomics anywhere in the pipeline. */
struct WfParams {
    uint32_t W, H, firstFrame, numFrames;   /* frames of this batch */
    TbTileMap tiles;
    float4* samples;                         /* [numFrames][W*H]: (o0, o1, o2, +-o3), sign of .w = jitter coin < 0.5 */
    uint32_t segCapacity, numSegments;
    unsigned long long* prof;                /* pipeline 3, counting launch: WaveProf slots (pt_device.hpp), else null */
    uint32_t pathsPerLane;                   /* pipeline 3 (pt_pooled.inc): samples of its pixel a lane keeps in flight, 1 or 2 */
    uint32_t sortByMaterial;                 /* wf_shade: shade the entries of a segment in material order (counting sort of indices in LDS) */
    uint32_t refillBelow;                    /* wf_extend: > 0 = the persistent form, a wave claims new entries when this many of its lanes are idle */
};

#define WF_STAGE_GENERATE_EXTEND 0
#define WF_STAGE_SHADE 1
#define WF_STAGE_CONNECT 2
#define WF_STAGE_EXTEND 3
#define WF_STAGE_ACCUMULATE 4
#define WF_STAGE_POOLED 5          /* pipeline 3: the whole batch in one persistent launch with an LDS ray pool */
