//! aruco3_hip.rs -- the Rust side of the drop-in boundary: binds the aruco3 crate to libaruco3_hip.so
//! (include/aruco3_hip.h), the MI355X implementation of `Detector::detect` and the pose solvers.
//!
//! How a maintainer adds it (INTEGRATION.md has the step-by-step):
//!   * copy this file to `src/hip.rs`, add `mod hip;` to `src/lib.rs` (next to `mod aruco;`, src/lib.rs:1-4);
//!   * delete the body of `impl Detector` in src/aruco.rs:51-122 and of the three public solvers in
//!     src/pose.rs:52-81 -- their replacements are the `impl Detector` block and `pub mod pose_hip` below;
//!   * `build.rs` links `aruco3_hip`.
//!
//! What does NOT change: `Detector { config, dictionary }` keeps exactly its two public fields
//! (src/aruco.rs:46-49), so every struct literal in the wild -- benches/detect_markers.rs:17,33,
//! examples/webcam_kamera.rs:14, examples/macroquad_detect.rs:18, README.md:14 -- compiles unchanged;
//! `detect(&self, image: DynamicImage) -> Detection` keeps its signature (src/aruco.rs:52) and its
//! error style (panic).  The device context (`a3_ctx`) therefore cannot live inside the struct: it
//! lives in a process-wide registry keyed by the detector's VALUE (config bits + dictionary identity),
//! see `registry()`.
//!
//! Status: written against include/aruco3_hip.h ABI version 3; NOT compiled in the build image (no
//! Rust toolchain there).  tests/test_rust_shim.py checks that every entry point of the header is
//! declared here with the same number of parameters and that the `#[repr(C)]` structs list the
//! header's fields in order.

#![allow(dead_code)]
#![allow(clippy::too_many_arguments)]

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::atomic::{AtomicBool, AtomicI32, Ordering};
use std::sync::{Arc, Mutex, OnceLock};

use image::{DynamicImage, GrayImage};
use imageproc::point::Point;
use nalgebra as na;

use crate::aruco::{Detection, Detector, DetectorConfig, Marker};
use crate::dictionaries::ARDictionary;
use crate::pinhole::CameraIntrinsics;
use crate::pose::MarkerPose;

// =====================================================================================================
// 1. The C ABI, one declaration per entry point of include/aruco3_hip.h
// =====================================================================================================

pub const A3_ABI_VERSION: c_int = 5;

pub const A3_OK: c_int = 0;
pub const A3_ERR_INVALID: c_int = -1;
pub const A3_ERR_HIP: c_int = -2;
pub const A3_ERR_CAPACITY: c_int = -3;
pub const A3_ERR_LIMIT: c_int = -6; // a fixed limit of the library (65536 candidates per frame): growing the output does not help
pub const A3_ERR_INTERNAL: c_int = -4;
pub const A3_ERR_NO_DEVICE: c_int = -5;

pub const A3_FMT_RGB8: c_int = 0;
pub const A3_FMT_RGBA8: c_int = 1;
pub const A3_FMT_L8: c_int = 2;
/// webcam byte order: examples/webcam_kamera.rs:38-52 re-orders it on the CPU today; the kernel reads it as it is
pub const A3_FMT_BGRA8: c_int = 3;
pub const A3_MEM_HOST: c_int = 0;
pub const A3_MEM_DEVICE: c_int = 1;

pub const A3_STAGE_THRESHOLD: c_int = 0;
pub const A3_STAGE_CONTOUR: c_int = 1;
pub const A3_STAGE_DECODE: c_int = 2;
pub const A3_PROFILE_OFF: c_int = 0;
pub const A3_PROFILE_STAGES: c_int = 1;
pub const A3_PROFILE_THRESHOLD_ONLY: c_int = 2;
pub const A3_PROFILE_THRESHOLD_SAMPLED: c_int = 3;

/// a3_config <-> DetectorConfig, src/aruco.rs:23-30
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct A3Config {
    pub threshold_window: u32,
    pub contour_simplification_epsilon: f64,
    pub min_side_length_factor: f32,
    pub min_corner_separation_factor: f32,
    pub homography_sample_size: u32, // usize in the reference (src/aruco.rs:28): narrowed with a check in `to_a3_config`
    pub filter_high_bit_errors: u8,
}

/// a3_marker <-> Marker, src/aruco.rs:8-13 (+ the frame it belongs to in a batch)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct A3Marker {
    pub frame: u32,
    pub id: u32,
    pub code: u64,
    pub corners: [u32; 8],
    pub hamming_distance: u8,
    pub rotation: u8,
    pub candidate_index: u16,
}

/// a3_pose <-> MarkerPose, src/pose.rs:8-12; rotation row-major, as `Matrix3::new(..)` is written
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct A3Pose {
    pub error: f32,
    pub rotation: [f32; 9],
    pub translation: [f32; 3],
}

/// a3_intrinsics <-> CameraIntrinsics, src/pinhole.rs:11-18
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct A3Intrinsics {
    pub image_width: u32,
    pub image_height: u32,
    pub focal_x: f32,
    pub focal_y: f32,
    pub principal_x: f32,
    pub principal_y: f32,
}

/// a3_stats: per-batch stage counters (the reference prints its rejects in debug builds, src/aruco.rs:163-164)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct A3Stats {
    pub darts: u64,
    pub contours_traced: u64,
    pub contours_materialised: u64,
    pub candidates_pre: u64,
    pub candidates: u64,
    pub markers: u64,
    pub resolve_iterations: u32,
    pub jump_rounds: u32,
    pub chunks: u32,
    /// bits 0-7: A3_STEP_* (how the library scheduled the batch: 0 whole, 1 decode deferred, 2 chain held and released by the burst's
    /// last member, 3 chain held and released early, 4 the burst's last member, 5 chain still held: between submit and collect only); bits 8-15: other contexts' chains this submit released;
    /// bits 16-23: synchronous re-runs of the batch the device asked for
    pub stepping: u32,
}

/// a3_synth_marker / a3_synth_frame: layouts for the device-side synthetic frame generator
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct A3SynthMarker {
    pub hinv: [f32; 9],
    pub x0: i32,
    pub y0: i32,
    pub x1: i32,
    pub y1: i32,
    pub cells: [u64; 2],
    pub n: u32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct A3SynthFrame {
    pub base: f32,
    pub gx: f32,
    pub gy: f32,
    pub noise_sigma: f32,
    pub first_marker: u32,
    pub n_markers: u32,
    pub seed: u64,
}

/// opaque a3_ctx
#[repr(C)]
pub struct A3Ctx {
    _private: [u8; 0],
}

extern "C" {
    pub fn a3_abi_version() -> c_int;
    pub fn a3_default_config(cfg: *mut A3Config);
    pub fn a3_create(device: c_int, cfg: *const A3Config, codes: *const u64, n_codes: usize, num_bits: u8, tau: u8,
                     out: *mut *mut A3Ctx) -> c_int;
    pub fn a3_destroy(ctx: *mut A3Ctx);
    pub fn a3_last_error(ctx: *const A3Ctx) -> *const c_char;
    pub fn a3_set_stream(ctx: *mut A3Ctx, hip_stream: *mut c_void) -> c_int;
    pub fn a3_get_stream(ctx: *const A3Ctx, hip_stream: *mut *mut c_void) -> c_int;
    pub fn a3_set_pool_limits(ctx: *mut A3Ctx, max_darts: u64, max_points: u64) -> c_int;
    pub fn a3_get_tau(ctx: *const A3Ctx, tau: *mut u8) -> c_int;
    pub fn a3_set_debug_taps(ctx: *mut A3Ctx, enabled: c_int) -> c_int;
    pub fn a3_order_after(ctx: *mut A3Ctx, other: *mut A3Ctx) -> c_int;
    pub fn a3_detect_batch(ctx: *mut A3Ctx, pixels: *const c_void, memory: c_int, fmt: c_int, width: u32, height: u32,
                           row_stride: usize, frame_stride: usize, n_frames: u32, out: *mut A3Marker, out_cap: usize,
                           per_frame_count: *mut u32, out_n: *mut usize) -> c_int;
    pub fn a3_detect_batch_pose(ctx: *mut A3Ctx, pixels: *const c_void, memory: c_int, fmt: c_int, width: u32, height: u32,
                                row_stride: usize, frame_stride: usize, n_frames: u32, marker_size_mm: f32,
                                intr: *const A3Intrinsics, out: *mut A3Marker, poses: *mut A3Pose, out_cap: usize,
                                per_frame_count: *mut u32, out_n: *mut usize) -> c_int;
    pub fn a3_detect_batch_submit(ctx: *mut A3Ctx, pixels: *const c_void, memory: c_int, fmt: c_int, width: u32, height: u32,
                                  row_stride: usize, frame_stride: usize, n_frames: u32, out_cap: usize) -> c_int;
    pub fn a3_detect_batch_collect(ctx: *mut A3Ctx, out: *mut A3Marker, out_cap: usize, per_frame_count: *mut u32,
                                   out_n: *mut usize) -> c_int;
    pub fn a3_detect_batch_pose_submit(ctx: *mut A3Ctx, pixels: *const c_void, memory: c_int, fmt: c_int, width: u32, height: u32,
                                       row_stride: usize, frame_stride: usize, n_frames: u32, marker_size_mm: f32,
                                       intr: *const A3Intrinsics, out_cap: usize) -> c_int;
    pub fn a3_detect_batch_pose_collect(ctx: *mut A3Ctx, out: *mut A3Marker, poses: *mut A3Pose, out_cap: usize,
                                        per_frame_count: *mut u32, out_n: *mut usize) -> c_int;
    pub fn a3_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn a3_host_free(p: *mut c_void) -> c_int;
    pub fn a3_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn a3_host_unregister(p: *mut c_void) -> c_int;
    pub fn a3_get_stats(ctx: *const A3Ctx, stats: *mut A3Stats) -> c_int;
    pub fn a3_download_grey(ctx: *mut A3Ctx, frame: u32, dst: *mut u8) -> c_int;
    pub fn a3_download_thresholded(ctx: *mut A3Ctx, frame: u32, dst: *mut u8) -> c_int;
    pub fn a3_candidate_count(ctx: *mut A3Ctx, frame: u32, n_pre: *mut u32, n_final: *mut u32) -> c_int;
    pub fn a3_download_candidates(ctx: *mut A3Ctx, frame: u32, before_discard: c_int, dst_xy: *mut u32, cap_quads: usize) -> c_int;
    pub fn a3_download_homographies(ctx: *mut A3Ctx, frame: u32, dst: *mut u8, ok: *mut u8, codes4: *mut u64,
                                    decode_ok: *mut i32, cap: usize) -> c_int;
    pub fn a3_contour_count(ctx: *mut A3Ctx, frame: u32, n_contours: *mut u32, n_points: *mut u64) -> c_int;
    pub fn a3_download_contours(ctx: *mut A3Ctx, frame: u32, start_keys: *mut u32, lengths: *mut u32, points_xy: *mut u32,
                                cap_contours: usize, cap_points: usize) -> c_int;
    pub fn a3_detection_record_bytes(max_markers_per_frame: u32, with_poses: c_int) -> usize;
    pub fn a3_pack_detections(ctx: *mut A3Ctx, first_frame_global: u32, max_markers_per_frame: u32, with_poses: c_int,
                              dst_device: *mut c_void, dst_bytes: usize) -> c_int;
    pub fn a3_estimate_pose(ctx: *mut A3Ctx, corners_xy: *const u32, n: usize, marker_size_mm: f32, intr: *const A3Intrinsics,
                            image_width: u32, image_height: u32, out: *mut A3Pose) -> c_int;
    pub fn a3_estimate_pose_normalized(ctx: *mut A3Ctx, points_xy: *const f32, n: usize, marker_size_mm: f32,
                                       out: *mut A3Pose) -> c_int;
    pub fn a3_find_nearest(ctx: *mut A3Ctx, bits: *const u64, n: usize, idx: *mut u32, dist: *mut u8) -> c_int;
    pub fn a3_calculate_tau(device: c_int, codes: *const u64, n_codes: usize, tau: *mut u8) -> c_int;
    pub fn a3_set_profiling(ctx: *mut A3Ctx, mode: c_int) -> c_int;
    pub fn a3_get_profile(ctx: *mut A3Ctx, stage: c_int, total_ms: *mut f64, launches: *mut u64, reset: c_int) -> c_int;
    pub fn a3_synth_render(device: c_int, hip_stream: *mut c_void, frames: *const A3SynthFrame, n_frames: u32,
                           markers: *const A3SynthMarker, n_markers: u32, width: u32, height: u32, paper: c_int, black: f32,
                           white: f32, supersample: c_int, out_rgb_device: *mut c_void, row_stride: usize,
                           frame_stride: usize) -> c_int;
}

// =====================================================================================================
// 2. Contexts: owned by a process-wide registry, never by `Detector`
// =====================================================================================================

/// One `a3_ctx`.  A context is not re-entrant (header: "one a3_ctx per (device, stream)"), so it is only ever touched
/// through the `Mutex` of its registry slot; that keeps `&Detector` `Sync`, as the reference's plain struct is.
pub struct HipCtx {
    raw: *mut A3Ctx,
    /// pinned staging for the frames of a call, grow-only and re-used from call to call (page-locking memory costs far more
    /// than a frame's copy: it is not done per `detect()`)
    staging: PinnedBytes,
}
// the raw pointer is only dereferenced by the library, under the slot's Mutex
unsafe impl Send for HipCtx {}

impl HipCtx {
    fn create(device: c_int, cfg: &A3Config, dict: &ARDictionary) -> HipCtx {
        let mut raw: *mut A3Ctx = std::ptr::null_mut();
        // tau == 0 would ask the library to compute it (src/dictionaries.rs:124); ARDictionary::new_from_ar_dictionary has
        // already done so, the value is simply passed on
        let rc = unsafe { a3_create(device, cfg, dict.code_list.as_ptr(), dict.code_list.len(), dict.num_bits, dict.tau, &mut raw) };
        if rc != A3_OK {
            // the reference's error style: threshold_window == 0 trips an assert inside imageproc, a bad dictionary panics
            // (src/dictionaries.rs:144); a missing GPU has no precedent and panics with the library's message
            panic!("aruco3_hip: a3_create failed ({}): {}", rc, last_error(std::ptr::null()));
        }
        HipCtx { raw, staging: PinnedBytes::empty() }
    }
    fn check(&self, rc: c_int, what: &str) {
        if rc != A3_OK {
            panic!("aruco3_hip: {} failed ({}): {}", what, rc, last_error(self.raw));
        }
    }
}

impl Drop for HipCtx {
    fn drop(&mut self) {
        if !self.raw.is_null() {
            unsafe { a3_destroy(self.raw) };
            self.raw = std::ptr::null_mut();
        }
    }
}

fn last_error(ctx: *const A3Ctx) -> String {
    // ctx may be NULL: the message of the last failed a3_create
    let p = unsafe { a3_last_error(ctx) };
    if p.is_null() {
        return String::new();
    }
    unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
}

/// What makes two `Detector` values the same detector: every config field bit for bit, the dictionary's table
/// (`&'static [u64]`: address + length identify it for the life of the process), num_bits, tau, and the device.
#[derive(Clone, Copy, PartialEq, Eq, Hash, Debug)]
struct Key {
    threshold_window: u32,
    eps_bits: u64,
    min_side_bits: u32,
    min_sep_bits: u32,
    sample: usize,
    filter: bool,
    codes_ptr: usize,
    codes_len: usize,
    num_bits: u8,
    tau: u8,
    device: c_int,
}

fn key_of(d: &Detector, device: c_int) -> Key {
    Key {
        threshold_window: d.config.threshold_window,
        eps_bits: d.config.contour_simplification_epsilon.to_bits(),
        min_side_bits: d.config.min_side_length_factor.to_bits(),
        min_sep_bits: d.config.min_corner_separation_factor.to_bits(),
        sample: d.config.homography_sample_size,
        filter: d.config.filter_high_bit_errors,
        codes_ptr: d.dictionary.code_list.as_ptr() as usize,
        codes_len: d.dictionary.code_list.len(),
        num_bits: d.dictionary.num_bits,
        tau: d.dictionary.tau,
        device,
    }
}

fn to_a3_config(c: &DetectorConfig) -> A3Config {
    // usize -> u32 (src/aruco.rs:28 vs a3_config): the library accepts 1..=200, which covers every marker family
    // (the patch only has to hold the mark_size x mark_size grid); anything else is a caller error
    if c.homography_sample_size == 0 || c.homography_sample_size > 200 {
        panic!("aruco3_hip: homography_sample_size must be in 1..=200, got {}", c.homography_sample_size);
    }
    A3Config {
        threshold_window: c.threshold_window,
        contour_simplification_epsilon: c.contour_simplification_epsilon,
        min_side_length_factor: c.min_side_length_factor,
        min_corner_separation_factor: c.min_corner_separation_factor,
        homography_sample_size: c.homography_sample_size as u32,
        filter_high_bit_errors: c.filter_high_bit_errors as u8,
    }
}

type Slot = Arc<Mutex<HipCtx>>;
struct Registry {
    slots: HashMap<Key, (Slot, u64)>, // context + last-use tick
    tick: u64,
}
/// more distinct detectors than this alive at once evicts the least recently used context (each owns device buffers)
const MAX_CONTEXTS: usize = 8;

fn registry() -> &'static Mutex<Registry> {
    static REG: OnceLock<Mutex<Registry>> = OnceLock::new();
    REG.get_or_init(|| Mutex::new(Registry { slots: HashMap::new(), tick: 0 }))
}

static DEVICE: AtomicI32 = AtomicI32::new(0);
static POPULATE: AtomicBool = AtomicBool::new(true);

/// HIP device the contexts are created on (default 0).  One process per GPU is the intended deployment.
pub fn set_device(device: i32) {
    DEVICE.store(device, Ordering::Relaxed);
}

/// `Detection.grey`, `.candidates` and `.homographies` (src/aruco.rs:115-120) are filled by default, exactly as the
/// reference returns them.  They cost a 2 MB read-back per 1080p frame and a grey plane K1 otherwise never writes;
/// callers that only read `.markers` (both webcam examples) switch them off here and get `grey: None`, empty vectors.
pub fn set_populate_debug_outputs(on: bool) {
    POPULATE.store(on, Ordering::Relaxed);
}

/// Destroys every context (device buffers, stream).  Optional: contexts otherwise live until the process exits.
pub fn shutdown() {
    registry().lock().unwrap().slots.clear();
}

fn slot_for(d: &Detector) -> Slot {
    let device = DEVICE.load(Ordering::Relaxed);
    let key = key_of(d, device);
    let mut reg = registry().lock().unwrap();
    reg.tick += 1;
    let tick = reg.tick;
    if let Some((slot, used)) = reg.slots.get_mut(&key) {
        *used = tick;
        return slot.clone();
    }
    if reg.slots.len() >= MAX_CONTEXTS {
        if let Some(oldest) = reg.slots.iter().min_by_key(|(_, (_, used))| *used).map(|(k, _)| *k) {
            reg.slots.remove(&oldest); // dropped (a3_destroy) once no call holds its Arc any more
        }
    }
    let slot: Slot = Arc::new(Mutex::new(HipCtx::create(device, &to_a3_config(&d.config), &d.dictionary)));
    reg.slots.insert(key, (slot.clone(), tick));
    slot
}

// =====================================================================================================
// 3. Detector::detect (replaces src/aruco.rs:52-121) + the batch form the GPU wants
// =====================================================================================================

/// Page-locked host memory from `a3_host_alloc`: frames packed here cross the link asynchronously and at its full rate
/// (include/aruco3_hip.h, "Host frames").  Falls back to nothing: an allocation failure panics like every other error.
struct PinnedBytes {
    ptr: *mut u8,
    len: usize,
    cap: usize,
}
// only ever touched under the Mutex of the context that owns it
unsafe impl Send for PinnedBytes {}
impl PinnedBytes {
    fn empty() -> PinnedBytes {
        PinnedBytes { ptr: std::ptr::null_mut(), len: 0, cap: 0 }
    }
    /// empty the buffer and make room for `cap` bytes (re-allocating only when it has to grow)
    fn reset(&mut self, cap: usize) {
        self.len = 0;
        if cap <= self.cap {
            return;
        }
        if !self.ptr.is_null() {
            unsafe { a3_host_free(self.ptr as *mut c_void) };
            self.ptr = std::ptr::null_mut();
            self.cap = 0;
        }
        let mut p: *mut c_void = std::ptr::null_mut();
        let rc = unsafe { a3_host_alloc(cap, &mut p) };
        if rc != A3_OK || p.is_null() {
            panic!("aruco3_hip: a3_host_alloc({}) failed ({}): {}", cap, rc, last_error(std::ptr::null()));
        }
        self.ptr = p as *mut u8;
        self.cap = cap;
    }
    fn extend_from_slice(&mut self, src: &[u8]) {
        assert!(self.len + src.len() <= self.cap, "aruco3_hip: packed batch larger than computed");
        unsafe { std::ptr::copy_nonoverlapping(src.as_ptr(), self.ptr.add(self.len), src.len()) };
        self.len += src.len();
    }
    fn as_ptr(&self) -> *const u8 {
        self.ptr
    }
}
impl Drop for PinnedBytes {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { a3_host_free(self.ptr as *mut c_void) };
            self.ptr = std::ptr::null_mut();
        }
    }
}

/// One batch as the library wants it: frames of one size and one layout, back to back, in the context's pinned staging buffer.
struct Packed {
    fmt: c_int,
    bpp: usize,
    width: u32,
    height: u32,
    bytes: *const u8,
}

/// `DynamicImage` -> raw bytes without touching pixel values.  Rgb8 / Rgba8 / Luma8 buffers are handed over as they are
/// (the kernel applies `into_luma8`'s integer formula itself, src/aruco.rs:60); every other variant (16-bit, float,
/// LumaA) goes through the crate's own `into_luma8()` on the CPU first, so its result is the reference's by construction.
fn pack(images: &[DynamicImage], staging: &mut PinnedBytes) -> Packed {
    assert!(!images.is_empty(), "aruco3_hip: empty batch");
    let (width, height) = (images[0].width(), images[0].height());
    let kind = |im: &DynamicImage| -> (c_int, usize) {
        match im {
            DynamicImage::ImageRgb8(_) => (A3_FMT_RGB8, 3),
            DynamicImage::ImageRgba8(_) => (A3_FMT_RGBA8, 4),
            _ => (A3_FMT_L8, 1),
        }
    };
    let (fmt, bpp) = kind(&images[0]);
    let uniform = images.iter().all(|im| kind(im).0 == fmt);
    let (fmt, bpp) = if uniform { (fmt, bpp) } else { (A3_FMT_L8, 1) }; // mixed layouts: everything to Luma8
    staging.reset(images.len() * width as usize * height as usize * bpp);
    let bytes = staging;
    for im in images {
        assert!(im.width() == width && im.height() == height, "aruco3_hip: all frames of a batch must have one size");
        match (fmt, im) {
            (A3_FMT_RGB8, DynamicImage::ImageRgb8(b)) => bytes.extend_from_slice(b.as_raw()),
            (A3_FMT_RGBA8, DynamicImage::ImageRgba8(b)) => bytes.extend_from_slice(b.as_raw()),
            (_, DynamicImage::ImageLuma8(b)) => bytes.extend_from_slice(b.as_raw()),
            (_, other) => bytes.extend_from_slice(other.clone().into_luma8().as_raw()),
        }
    }
    Packed { fmt, bpp, width, height, bytes: bytes.as_ptr() }
}

fn marker_of(m: &A3Marker) -> Marker {
    Marker {
        id: m.id as usize,
        code: m.code,
        corners: (0..4).map(|i| (m.corners[2 * i], m.corners[2 * i + 1])).collect(),
        hamming_distance: m.hamming_distance,
    }
}

/// Detection.grey / .candidates / .homographies of frame `f` of the batch that just ran with debug taps on
fn fill_debug_outputs(ctx: &HipCtx, f: u32, width: u32, height: u32, sample: u32, det: &mut Detection) {
    let mut grey = vec![0u8; width as usize * height as usize];
    ctx.check(unsafe { a3_download_grey(ctx.raw, f, grey.as_mut_ptr()) }, "a3_download_grey");
    det.grey = GrayImage::from_raw(width, height, grey);
    let (mut n_pre, mut n_final) = (0u32, 0u32);
    ctx.check(unsafe { a3_candidate_count(ctx.raw, f, &mut n_pre, &mut n_final) }, "a3_candidate_count");
    let n = n_final as usize;
    let mut xy = vec![0u32; 8 * n.max(1)];
    ctx.check(unsafe { a3_download_candidates(ctx.raw, f, 0, xy.as_mut_ptr(), n.max(1)) }, "a3_download_candidates");
    det.candidates = (0..n).map(|k| (0..4).map(|i| Point::new(xy[8 * k + 2 * i], xy[8 * k + 2 * i + 1])).collect()).collect();
    let s2 = (sample * sample) as usize;
    let mut patches = vec![0u8; s2 * n.max(1)];
    let mut ok = vec![0u8; n.max(1)];
    ctx.check(unsafe { a3_download_homographies(ctx.raw, f, patches.as_mut_ptr(), ok.as_mut_ptr(), std::ptr::null_mut(),
                                                std::ptr::null_mut(), n.max(1)) }, "a3_download_homographies");
    det.homographies = (0..n).map(|k| {
        if ok[k] != 0 {
            GrayImage::from_raw(sample, sample, patches[k * s2..(k + 1) * s2].to_vec()).unwrap()
        } else {
            GrayImage::new(1, 1) // src/aruco.rs:256: a failed projection leaves a 1x1 black image
        }
    }).collect();
}

/// With `Detection.homographies` populated the library keeps one patch per candidate of the batch, up to 2^20 patches per call;
/// a populated call therefore never carries more frames than this (larger batches are split).
const MAX_TAPPED_FRAMES: usize = 1024;
/// The library's candidate tables grow to 65536 quads per frame (beyond that a3_detect_batch reports A3_ERR_LIMIT, which no retry
/// cures and which `check` turns into a panic at once): marker lists are grown and the batch re-run only on A3_ERR_CAPACITY, and at
/// most up to that many markers per frame.
const MAX_MARKERS_PER_FRAME: usize = 65536;

impl Detector {
    /// src/aruco.rs:52-121, same signature.  One frame = a batch of one.
    pub fn detect(&self, image: DynamicImage) -> Detection {
        self.detect_batch(std::slice::from_ref(&image)).pop().unwrap()
    }

    /// New (additive): many independent frames per call, all of one size -- what the GPU wants.  Results are in frame order;
    /// each `Detection` is what `detect` would have returned for that frame.
    pub fn detect_batch(&self, images: &[DynamicImage]) -> Vec<Detection> {
        if images.is_empty() {
            return Vec::new();
        }
        let populate = POPULATE.load(Ordering::Relaxed);
        if populate && images.len() > MAX_TAPPED_FRAMES {
            return images.chunks(MAX_TAPPED_FRAMES).flat_map(|c| self.detect_batch(c)).collect();
        }
        let slot = slot_for(self);
        let mut ctx = slot.lock().unwrap();
        let p = pack(images, &mut ctx.staging);
        ctx.check(unsafe { a3_set_debug_taps(ctx.raw, populate as c_int) }, "a3_set_debug_taps");
        let n = images.len();
        let mut markers = vec![A3Marker::default(); 64 * n];
        let mut per = vec![0u32; n];
        let mut found = 0usize;
        loop {
            let rc = unsafe {
                a3_detect_batch(ctx.raw, p.bytes as *const c_void, A3_MEM_HOST, p.fmt, p.width, p.height,
                                p.width as usize * p.bpp, p.width as usize * p.height as usize * p.bpp, n as u32,
                                markers.as_mut_ptr(), markers.len(), per.as_mut_ptr(), &mut found)
            };
            // the reference has no marker limit: a list that does not fit is grown and the batch re-run
            if rc == A3_ERR_CAPACITY && markers.len() < MAX_MARKERS_PER_FRAME * n {
                markers.resize(markers.len() * 4, A3Marker::default());
                continue;
            }
            ctx.check(rc, "a3_detect_batch");
            break;
        }
        let mut out = Vec::with_capacity(n);
        let mut pos = 0usize;
        for f in 0..n {
            let cnt = per[f] as usize;
            let mut det = Detection { grey: None, candidates: vec![], homographies: vec![], markers: markers[pos..pos + cnt].iter().map(marker_of).collect() };
            pos += cnt;
            if populate {
                fill_debug_outputs(&ctx, f as u32, p.width, p.height, self.config.homography_sample_size as u32, &mut det);
            }
            out.push(det);
        }
        out
    }

    /// New (additive): detect + both IPPE poses of every marker in one device pass (callers always solve the pose right after
    /// detect: examples/webcam_kamera.rs:56-71).  `intrinsics == None`: `solve_with_undistorted_points` with the frame size.
    pub fn detect_batch_with_pose(&self, images: &[DynamicImage], marker_size_mm: f32, intrinsics: Option<&CameraIntrinsics>)
        -> Vec<(Detection, Vec<(MarkerPose, MarkerPose)>)> {
        if images.is_empty() {
            return Vec::new();
        }
        let slot = slot_for(self);
        let mut ctx = slot.lock().unwrap();
        let p = pack(images, &mut ctx.staging);
        ctx.check(unsafe { a3_set_debug_taps(ctx.raw, 0) }, "a3_set_debug_taps");
        let n = images.len();
        let mut cap = 64 * n;
        let mut markers = vec![A3Marker::default(); cap];
        let mut poses = vec![A3Pose::default(); 2 * cap];
        let mut per = vec![0u32; n];
        let mut found = 0usize;
        let intr = intrinsics.map(to_a3_intrinsics);
        let intr_ptr = intr.as_ref().map_or(std::ptr::null(), |i| i as *const A3Intrinsics);
        loop {
            let rc = unsafe {
                a3_detect_batch_pose(ctx.raw, p.bytes as *const c_void, A3_MEM_HOST, p.fmt, p.width, p.height,
                                     p.width as usize * p.bpp, p.width as usize * p.height as usize * p.bpp, n as u32, marker_size_mm,
                                     intr_ptr, markers.as_mut_ptr(), poses.as_mut_ptr(), cap, per.as_mut_ptr(), &mut found)
            };
            // the reference has no marker limit: lists that do not fit are grown and the batch re-run (as in detect_batch)
            if rc == A3_ERR_CAPACITY && cap < MAX_MARKERS_PER_FRAME * n {
                cap *= 4;
                markers.resize(cap, A3Marker::default());
                poses.resize(2 * cap, A3Pose::default());
                continue;
            }
            ctx.check(rc, "a3_detect_batch_pose");
            break;
        }
        let mut out = Vec::with_capacity(n);
        let mut pos = 0usize;
        for f in 0..n {
            let cnt = per[f] as usize;
            let det = Detection { grey: None, candidates: vec![], homographies: vec![], markers: markers[pos..pos + cnt].iter().map(marker_of).collect() };
            let pp = (pos..pos + cnt).map(|i| (to_marker_pose(&poses[2 * i]), to_marker_pose(&poses[2 * i + 1]))).collect();
            pos += cnt;
            out.push((det, pp));
        }
        out
    }
}

// =====================================================================================================
// 3b. BatchQueue: several batches in flight, stepped in bursts (include/aruco3_hip.h, a3_order_after)
// =====================================================================================================

/// New (additive).  A capture / processing loop that must keep the GPU fed keeps `depth` batches in flight: `submit` hands a
/// batch over and returns at once, `collect` waits for the OLDEST batch in flight and returns its detections (markers only:
/// `Detection.grey` / `.candidates` / `.homographies` stay empty).  Inside, `depth` contexts of their own (not the registry's) are
/// used in rotation, each on a stream of its own, with the burst gates the header describes: before each submit context k calls
/// `a3_order_after` for the contexts k+1 .. depth-1, so the threshold kernels of one rotation run back to back after the previous
/// rotation's chains have drained and the contour / decode chains of the rotation run together.  Measured with a batch of its own per
/// context (round 5) a free-running rotation -- the same calls without `a3_order_after` -- is as fast within
/// 2 %, and depth 3 or 4 within 1 % of each other; the gates are kept because they give a rotation a quiet threshold phase and a quiet
/// chain phase, which is what a caller wants who runs something else (a collective, a copy) beside the detector.  Results never depend
/// on any of it.
/// The burst stepping is what the library does with these calls by itself (include/aruco3_hip.h, a3_order_after: a context that
/// declared gates holds its chain back behind its threshold kernel, the rotation's last context releases them) -- this type calls
/// nothing outside the public header, and `last_stepping()` reports what the library did with the batch just collected.
/// Export GPU_MAX_HW_QUEUES=8 before the process touches HIP (header, "Hardware queues").
pub struct BatchQueue {
    ctxs: Vec<HipCtx>,
    frames_in: Vec<usize>, // frames of the batch in flight on each context
    shape_in: Vec<(c_int, u32, u32, usize)>, // (fmt, width, height, bytes per pixel) of that batch, for the re-run of a marker-dense one
    last_stepping: u32,
    head: usize,           // context of the oldest batch in flight
    in_flight: usize,
    submitted: usize,
}

impl BatchQueue {
    pub fn new(detector: &Detector, depth: usize) -> BatchQueue {
        assert!(depth >= 1 && depth <= 8, "aruco3_hip: BatchQueue depth must be in 1..=8");
        let device = DEVICE.load(Ordering::Relaxed);
        let cfg = to_a3_config(&detector.config);
        let ctxs: Vec<HipCtx> = (0..depth).map(|_| HipCtx::create(device, &cfg, &detector.dictionary)).collect();
        BatchQueue { frames_in: vec![0; depth], shape_in: vec![(0, 0, 0, 0); depth], last_stepping: 0, ctxs, head: 0, in_flight: 0, submitted: 0 }
    }

    /// batches in flight
    pub fn len(&self) -> usize {
        self.in_flight
    }
    pub fn is_full(&self) -> bool {
        self.in_flight == self.ctxs.len()
    }
    /// a3_stats.stepping of the batch `collect` returned last (A3_STEP_* in bits 0-7)
    pub fn last_stepping(&self) -> u32 {
        self.last_stepping
    }

    /// Enqueue one batch (all frames of one size).  Panics if `depth` batches are already in flight: collect first.
    pub fn submit(&mut self, images: &[DynamicImage]) {
        assert!(!self.is_full(), "aruco3_hip: BatchQueue is full: collect() the oldest batch first");
        assert!(!images.is_empty(), "aruco3_hip: empty batch");
        let depth = self.ctxs.len();
        let k = self.submitted % depth;
        for m in k + 1..depth {
            // burst gate: this batch's threshold kernel starts once the batches in flight on the later contexts (the previous
            // rotation's) have drained.  A scheduling hint: a3_order_after never changes a result.
            let (this, other) = (self.ctxs[k].raw, self.ctxs[m].raw);
            self.ctxs[k].check(unsafe { a3_order_after(this, other) }, "a3_order_after");
        }
        let ctx = &mut self.ctxs[k];
        let p = pack(images, &mut ctx.staging); // pinned staging owned by the context: stays untouched until collect
        ctx.check(unsafe { a3_set_debug_taps(ctx.raw, 0) }, "a3_set_debug_taps");
        let n = images.len();
        let rc = unsafe {
            a3_detect_batch_submit(ctx.raw, p.bytes as *const c_void, A3_MEM_HOST, p.fmt, p.width, p.height, p.width as usize * p.bpp,
                                   p.width as usize * p.height as usize * p.bpp, n as u32, MAX_MARKERS_PER_FRAME.min(64) * n)
        };
        ctx.check(rc, "a3_detect_batch_submit");
        self.frames_in[k] = n;
        self.shape_in[k] = (p.fmt, p.width, p.height, p.bpp);
        self.submitted += 1;
        self.in_flight += 1;
    }

    /// Wait for the oldest batch in flight; one `Detection` per frame, in frame order.
    pub fn collect(&mut self) -> Vec<Detection> {
        assert!(self.in_flight > 0, "aruco3_hip: BatchQueue::collect with nothing in flight");
        let k = self.head;
        let ctx = &self.ctxs[k];
        let n = self.frames_in[k];
        let mut markers = vec![A3Marker::default(); 64 * n];
        let mut per = vec![0u32; n];
        let mut found = 0usize;
        let mut rc = unsafe { a3_detect_batch_collect(ctx.raw, markers.as_mut_ptr(), markers.len(), per.as_mut_ptr(), &mut found) };
        // the batch has left the queue whatever collect said (the library keeps no batch in flight after a failed collect)
        self.head = (self.head + 1) % self.ctxs.len();
        self.in_flight -= 1;
        let mut stats = A3Stats::default();
        if unsafe { a3_get_stats(ctx.raw, &mut stats) } == A3_OK {
            self.last_stepping = stats.stepping;
        }
        // The reference's marker Vec is unbounded (src/aruco.rs:75-113).  A frame with more markers than the list submitted for is
        // A3_ERR_CAPACITY: the frames are still in the context's pinned staging, so the batch is run again synchronously with a
        // list that grows, exactly as Detector::detect_batch does.
        while rc == A3_ERR_CAPACITY && markers.len() < MAX_MARKERS_PER_FRAME * n {
            markers.resize(markers.len() * 4, A3Marker::default());
            let (fmt, width, height, bpp) = self.shape_in[k];
            rc = unsafe {
                a3_detect_batch(ctx.raw, ctx.staging.as_ptr() as *const c_void, A3_MEM_HOST, fmt, width, height, width as usize * bpp,
                                width as usize * height as usize * bpp, n as u32, markers.as_mut_ptr(), markers.len(), per.as_mut_ptr(), &mut found)
            };
        }
        ctx.check(rc, "a3_detect_batch_collect");
        let mut out = Vec::with_capacity(n);
        let mut pos = 0usize;
        for f in 0..n {
            let cnt = per[f] as usize;
            out.push(Detection { grey: None, candidates: vec![], homographies: vec![], markers: markers[pos..pos + cnt].iter().map(marker_of).collect() });
            pos += cnt;
        }
        out
    }
}

// =====================================================================================================
// 4. Pose: the bodies of src/pose.rs:52-81 + README.md:34's `estimate_pose`
// =====================================================================================================

fn to_a3_intrinsics(c: &CameraIntrinsics) -> A3Intrinsics {
    A3Intrinsics { image_width: c.image_width, image_height: c.image_height, focal_x: c.focal_x, focal_y: c.focal_y,
                   principal_x: c.principal_x, principal_y: c.principal_y }
}

fn to_marker_pose(p: &A3Pose) -> MarkerPose {
    let r = &p.rotation;
    MarkerPose {
        error: p.error,
        rotation: na::Matrix3::new(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]), // row-major, like Matrix3::new
        translation: na::Vector3::new(p.translation[0], p.translation[1], p.translation[2]),
    }
}

/// The pose solvers are free functions with no detector to hang a context on: they share one small context of their own
/// (default config, empty dictionary -- the pose kernel reads neither).
fn pose_slot() -> &'static Mutex<HipCtx> {
    static POSE: OnceLock<Mutex<HipCtx>> = OnceLock::new();
    POSE.get_or_init(|| {
        let mut cfg = A3Config { threshold_window: 0, contour_simplification_epsilon: 0.0, min_side_length_factor: 0.0,
                                 min_corner_separation_factor: 0.0, homography_sample_size: 0, filter_high_bit_errors: 0 };
        unsafe { a3_default_config(&mut cfg) };
        let mut raw: *mut A3Ctx = std::ptr::null_mut();
        let rc = unsafe { a3_create(DEVICE.load(Ordering::Relaxed), &cfg, std::ptr::null(), 0, 1, 1, &mut raw) };
        if rc != A3_OK {
            panic!("aruco3_hip: a3_create (pose context) failed ({}): {}", rc, last_error(std::ptr::null()));
        }
        Mutex::new(HipCtx { raw, staging: PinnedBytes::empty() })
    })
}

/// Drop-in bodies for `pub mod pose` (src/pose.rs:52-81): same names, same arguments, same return order (lower error first).
pub mod pose_hip {
    use super::*;

    pub fn solve_with_intrinsics(image_points: &Vec<(u32, u32)>, marker_size_mm: f32, camera_intrinsics: &CameraIntrinsics)
        -> (MarkerPose, MarkerPose) {
        solve_pixels(image_points, marker_size_mm, Some(camera_intrinsics), (camera_intrinsics.image_width, camera_intrinsics.image_height))
    }

    pub fn solve_with_undistorted_points(image_points: &Vec<(u32, u32)>, marker_size_mm: f32, image_size: (u32, u32))
        -> (MarkerPose, MarkerPose) {
        solve_pixels(image_points, marker_size_mm, None, image_size)
    }

    pub fn solve_with_normalized_points(normalized_image_points: &Vec<(f32, f32)>, marker_size_mm: f32) -> (MarkerPose, MarkerPose) {
        assert!(normalized_image_points.len() == 4, "aruco3_hip: a marker has 4 corners");
        let pts: Vec<f32> = normalized_image_points.iter().flat_map(|&(x, y)| [x, y]).collect();
        let mut out = [A3Pose::default(); 2];
        let ctx = pose_slot().lock().unwrap();
        ctx.check(unsafe { a3_estimate_pose_normalized(ctx.raw, pts.as_ptr(), 1, marker_size_mm, out.as_mut_ptr()) },
                  "a3_estimate_pose_normalized");
        (to_marker_pose(&out[0]), to_marker_pose(&out[1]))
    }

    fn solve_pixels(image_points: &Vec<(u32, u32)>, marker_size_mm: f32, intr: Option<&CameraIntrinsics>, image_size: (u32, u32))
        -> (MarkerPose, MarkerPose) {
        assert!(image_points.len() == 4, "aruco3_hip: a marker has 4 corners");
        let xy: Vec<u32> = image_points.iter().flat_map(|&(x, y)| [x, y]).collect();
        let a3i = intr.map(to_a3_intrinsics);
        let intr_ptr = a3i.as_ref().map_or(std::ptr::null(), |i| i as *const A3Intrinsics);
        let mut out = [A3Pose::default(); 2];
        let ctx = pose_slot().lock().unwrap();
        ctx.check(unsafe { a3_estimate_pose(ctx.raw, xy.as_ptr(), 1, marker_size_mm, intr_ptr, image_size.0, image_size.1, out.as_mut_ptr()) },
                  "a3_estimate_pose");
        (to_marker_pose(&out[0]), to_marker_pose(&out[1]))
    }
}

/// README.md:34 -- `estimate_pose((1920, 1080), &d.corners, MARKER_SIZE_IN_MM, None)`.  The README names it, the source
/// never defined it; re-export it from src/lib.rs next to `Detector`.
pub fn estimate_pose(image_size: (u32, u32), corners: &Vec<(u32, u32)>, marker_size_mm: f32, intrinsics: Option<&CameraIntrinsics>)
    -> (MarkerPose, MarkerPose) {
    match intrinsics {
        Some(i) => pose_hip::solve_with_intrinsics(corners, marker_size_mm, i),
        None => pose_hip::solve_with_undistorted_points(corners, marker_size_mm, image_size),
    }
}

/// `ARDictionary::find_nearest` for many codes at once on the device (src/dictionaries.rs:160-196); the single-code method of
/// the reference stays as it is (a 1023-entry scan is not worth a launch).
pub fn find_nearest_batch(d: &Detector, bits: &[u64]) -> Vec<(usize, u8)> {
    let slot = slot_for(d);
    let ctx = slot.lock().unwrap();
    let mut idx = vec![0u32; bits.len().max(1)];
    let mut dist = vec![0u8; bits.len().max(1)];
    ctx.check(unsafe { a3_find_nearest(ctx.raw, bits.as_ptr(), bits.len(), idx.as_mut_ptr(), dist.as_mut_ptr()) }, "a3_find_nearest");
    (0..bits.len()).map(|i| (idx[i] as usize, dist[i])).collect()
}

/// per-batch stage counters of the detector's last call
pub fn last_stats(d: &Detector) -> A3Stats {
    let slot = slot_for(d);
    let ctx = slot.lock().unwrap();
    let mut s = A3Stats::default();
    ctx.check(unsafe { a3_get_stats(ctx.raw, &mut s) }, "a3_get_stats");
    s
}
