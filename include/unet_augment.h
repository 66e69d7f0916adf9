/* C ABI of the on-GPU training-sample augmentation, exported by libunet_hip.so.
 *
 * Replaces visual_perception_augmentation_cuda (visual_perception_augmentation.cu:282-544) and its 12 kernels
 * (.cu:6-280): the image channels and the label volume are already in HBM, are augmented in place and stay there
 * (the reference uploads and downloads every sample, .cu:294-295,529-530).
 *
 * The reference interleaves random draws (tipl::uniform_dist, TIPL, not in the reference tree) with kernel launches.
 * Here the two are separated: the caller resolves EVERY random decision on the host, in the reference's draw order,
 * into one UnetAugmentRecipe (the reference-side binding fills it from its own tipl::uniform_dist / std::mt19937 and
 * passes the tipl::transformation_matrix values it already builds, .cu:402-414,459-468), and
 * unet_augment_run is a deterministic function of (recipe, image, label).  Volumes are x-fastest fp32:
 * image = `channels` volumes of dims[2]*dims[1]*dims[0] stacked along z (train.cpp / .cu:305-311), label = one volume.
 *
 * Status codes / errors as in unet_hip.h (0 = ok, the message is read with its unet_last_error).
 */
#ifndef UNET_AUGMENT_H
#define UNET_AUGMENT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_AUG_MAX_CHANNELS 8
#define UNET_AUG_MAX_FOCI 16      /* options.txt:31 allows up to 10 distortion foci */
#define UNET_AUG_STAMPS 5         /* .cu:461 */

/* pos' = sr * pos + shift, sr row-major: tipl::transformation_matrix<float>::operator() as used at .cu:187,470 */
typedef struct { float sr[9]; float shift[3]; } UnetAugAffine;

typedef struct {
    int dims[3];                  /* image_shape: width (x), height (y), depth (z) */
    int channels;                 /* input.depth() / image_shape[2], .cu:305 */
    int is_label;                 /* label volume holds class ids (majority resampling, background stage) or an image */

    /* .cu:315-331: resolution loss; low_dims = the low-resolution grid, used when downsample != 0 */
    int downsample;
    int low_dims[3];
    /* .cu:333-341 cropping_at (.cu:6-29): sphere of `crop_radius` voxels around crop_pos */
    int crop;
    int crop_pos[3];
    float crop_radius, crop_value;
    /* .cu:343-354 truncate_top / truncate_buttom (.cu:31-59): whole z-slices zeroed in label and image */
    int trunc_top, trunc_bottom;
    /* .cu:356-361 add_noise (.cu:63-77): + noise_mag * U(0,1] per voxel.  The reference seeds curand with 0 for every
       sample; the stream here is a counter hash of (noise_seed, voxel index) */
    int noise;
    float noise_mag;
    unsigned noise_seed;
    /* .cu:363-368 */
    int ambient;
    float ambient_value;
    /* .cu:369-374 diffuse_light_cuda (.cu:79-96): direction as drawn (not normalised), magnitude = options["diffuse_mag"] */
    int diffuse;
    float diffuse_dir[3], diffuse_mag;
    /* .cu:375-380 specular_light_cuda (.cu:99-116) */
    int specular;
    int specular_pos[3];
    float specular_freq, specular_mag;

    /* .cu:383-446 the geometric stage.  `view` = the tipl::transformation_matrix built at .cu:402 */
    UnetAugAffine view;
    int has_perspective;          /* options["perspective"] > 0, .cu:432 */
    float perspective[3];
    int has_lens;                 /* options["lens_distortion"] > 0: gates BOTH the lens field and the foci, .cu:169-170,414,432 */
    float lens_magnitude;         /* range(0,1) * options["lens_distortion"] */
    int n_foci;                   /* .cu:417-429 create_distortion_at_cuda (.cu:140-161) */
    int foci_pos[UNET_AUG_MAX_FOCI][3];
    float foci_radius[UNET_AUG_MAX_FOCI], foci_magnitude[UNET_AUG_MAX_FOCI];

    /* .cu:449-521 background stage (only when is_label) */
    int zero_background;          /* .cu:452-457: keep voxels with a label, stop */
    int rubber;                   /* .cu:460-488 */
    UnetAugAffine stamp[UNET_AUG_STAMPS];
    float stamp_mag[UNET_AUG_MAX_CHANNELS][UNET_AUG_STAMPS];   /* range(0,1)*options["rubber_stamping_mag"], drawn per channel per stamp */
    int perlin;                   /* .cu:490-513, kernels .cu:199-280 */
    unsigned char perm[512];      /* the shuffled table of .cu:492-495 */
    float perlin_zoom, perlin_mag;
} UnetAugmentRecipe;

/* Bytes of device scratch unet_augment_run needs for this recipe's dims / channels (resampled copies, reduction cells). */
int unet_augment_scratch_bytes(const UnetAugmentRecipe* recipe, size_t* bytes);

/* Augments in place: `image` (channels * D*H*W fp32) and `label` (D*H*W fp32) are device pointers; afterwards they hold
 * what the reference copies back at .cu:529-530.  Enqueued on `stream` (hipStream_t); no host synchronisation. */
int unet_augment_run(const UnetAugmentRecipe* recipe, float* image, float* label, void* scratch, size_t scratch_bytes,
                     void* stream);

/* ---- simulate_modality (train.cpp:43-117 with labels, :119-178 without; called per template sample at train.cpp:460-462) ----
 * A random polynomial contrast: tissue = per-label random level (or the image itself), smoothed twice; every voxel above 0.02
 * becomes pow(sum of 20 random monomials in (x, tissue, 1-x, 1-tissue), gamma); the result is stretched to [0,1] over the
 * labelled voxels (all voxels without labels).  As with the augmentation, the caller makes the draws (tipl::uniform_dist) in the
 * reference's order and passes them; the engine is a deterministic function of (recipe, t1w, label), in place on t1w. */
#define UNET_SIM_TERMS 20
#define UNET_SIM_MAX_LABELS 256

typedef struct {
    int dims[3];
    int with_label;                              /* 1: train.cpp:43 overload (label volume given), 0: train.cpp:119 overload */
    int max_label;                               /* label values are 0..max_label (< UNET_SIM_MAX_LABELS) */
    float lut[UNET_SIM_MAX_LABELS];              /* 0.4 + rand*0.2 per label value, train.cpp:56-58 */
    unsigned char term_a[UNET_SIM_TERMS], term_b[UNET_SIM_TERMS], term_c[UNET_SIM_TERMS], term_d[UNET_SIM_TERMS];   /* exponents 0..3 */
    float term_w[UNET_SIM_TERMS];                /* train.cpp:65-78 */
    float gamma;                                 /* 0.6 + 1.2*rand, train.cpp:80 */
} UnetSimulateRecipe;

int unet_simulate_modality_scratch_bytes(const UnetSimulateRecipe* recipe, size_t* bytes);
/* t1w: D*H*W fp32 in [0,1], rewritten in place; label: D*H*W fp32 label values (ignored when with_label == 0, may be NULL). */
int unet_simulate_modality_run(const UnetSimulateRecipe* recipe, float* t1w, const float* label, void* scratch, size_t scratch_bytes,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif
