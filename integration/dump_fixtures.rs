//! dump_fixtures.rs -- writes what the REFERENCE crate (and the `image` / `imageproc` releases its Cargo.lock resolves to)
//! computes for this repository's golden inputs, stage by stage, so that the CPU oracle of the HIP port can be pinned to the
//! real thing (the port's own image has no Rust toolchain; see INTEGRATION.md, "Pinning the image stages").
//!
//! Use (in a checkout of JosephCatrambone/aruco3, with this repository next to it):
//!
//!     cp <repo>/integration/dump_fixtures.rs tests/dump_fixtures.rs
//!     A3_FIXTURE_INPUTS=<repo>/tests/fixtures/inputs A3_FIXTURE_OUT=<repo>/tests/fixtures cargo test --release --test dump_fixtures -- --nocapture
//!     (cd <repo> && python -m pytest tests/test_reference_fixtures.py -q)
//!
//! It is an integration test of the crate: it sees the crate's public API (`Detector::detect`, `ARDictionary`) and the
//! crate's own dependencies (`image`, `imageproc`), nothing private.  The third-party stages are therefore called directly, in
//! the order and with the arguments `Detector::detect` uses (src/aruco.rs:60,61,64,133,143,244-253,264-273), on data taken from the
//! crate's own `Detection` where a private helper sits in between (the candidate quads after `discard_too_near`).
//!
//! NOT COMPILED in the port's repository (no cargo there).  Written against image 0.25 / imageproc 0.25; if a signature has moved
//! in the release your lock file picks, the fix is local to the line that calls it.
//!
//! Output: one `<name>.a3fx` per input.  Format (little-endian): the 6 bytes "A3FX1\n", then records until EOF:
//!     u32 name_len, name (UTF-8), u8 dtype (0 u8, 1 u32, 2 u64, 3 i32, 4 f32, 5 f64), u32 ndim, u32 dims[ndim], data
//! Records written per input (n = candidates, m = markers, c = contours, ms = mark size):
//!     versions            u8[len]      "image=<ver> imageproc=<ver>" if the build script provides them, else "unknown"
//!     grey                u8[h][w]     image.into_luma8()                                          src/aruco.rs:60
//!     thresholded         u8[h][w]     adaptive_threshold(&grey, 7)                                 :61
//!     contour_len         u32[c]       find_contours::<u32>(&thresholded): points per contour, in the order returned   :64
//!     contour_border      u8[c]        0 outer, 1 hole
//!     contour_points      u32[sum][2]  x, y of every point, contours concatenated
//!     dp_len / dp_points               approximate_polygon_dp(points, len * 0.05, true) per contour  :133
//!     hull_len / hull_points           convex_hull(dp) for the contours whose dp has exactly 4 points (0 for the others)   :143
//!     candidates          u32[n][4][2] Detection.candidates (after enforce_clockwise_corners and discard_too_near)       :67-69
//!     warp_src_xy         f32[n][49][49][2]  projection.invert() * (x, y) for every output pixel: where warp_into samples   :244-253
//!     projection_ok       u8[n]        from_control_points returned Some
//!     homographies        u8[n][49][49]  warp_into(grey, projection, Bilinear, 0) -- and Detection.homographies must equal it  :253
//!     homographies_equal_detection  u8[1]
//!     otsu                u8[n]        otsu_level(&homography)                                      :264
//!     binarized           u8[n][49][49]  threshold(&homography, otsu, Binary)                       :265
//!     resized             u8[n][ms][ms]  imageops::resize(&binarized, ms, ms, Triangle)              :273
//!     marker_id u32[m], marker_code u64[m], marker_corners u32[m][4][2], marker_hamming u8[m]       :75-113
use std::fs;
use std::io::Write;
use std::path::PathBuf;

use image::{DynamicImage, GrayImage, RgbImage, RgbaImage};
use imageproc::contours::BorderType;
use imageproc::geometric_transformations::{warp_into, Interpolation, Projection};
use imageproc::point::Point;

use aruco3::{ARDictionary, Detector, DetectorConfig};

struct Fx {
    buf: Vec<u8>,
}

impl Fx {
    fn new() -> Fx {
        Fx { buf: b"A3FX1\n".to_vec() }
    }
    fn head(&mut self, name: &str, dtype: u8, dims: &[usize]) {
        self.buf.extend_from_slice(&(name.len() as u32).to_le_bytes());
        self.buf.extend_from_slice(name.as_bytes());
        self.buf.push(dtype);
        self.buf.extend_from_slice(&(dims.len() as u32).to_le_bytes());
        for d in dims {
            self.buf.extend_from_slice(&(*d as u32).to_le_bytes());
        }
    }
    fn u8s(&mut self, name: &str, dims: &[usize], v: &[u8]) {
        assert_eq!(dims.iter().product::<usize>(), v.len(), "{}", name);
        self.head(name, 0, dims);
        self.buf.extend_from_slice(v);
    }
    fn u32s(&mut self, name: &str, dims: &[usize], v: &[u32]) {
        assert_eq!(dims.iter().product::<usize>(), v.len(), "{}", name);
        self.head(name, 1, dims);
        for x in v {
            self.buf.extend_from_slice(&x.to_le_bytes());
        }
    }
    fn u64s(&mut self, name: &str, dims: &[usize], v: &[u64]) {
        assert_eq!(dims.iter().product::<usize>(), v.len(), "{}", name);
        self.head(name, 2, dims);
        for x in v {
            self.buf.extend_from_slice(&x.to_le_bytes());
        }
    }
    fn f32s(&mut self, name: &str, dims: &[usize], v: &[f32]) {
        assert_eq!(dims.iter().product::<usize>(), v.len(), "{}", name);
        self.head(name, 4, dims);
        for x in v {
            self.buf.extend_from_slice(&x.to_le_bytes());
        }
    }
}

fn points_flat(list: &[Vec<Point<u32>>]) -> (Vec<u32>, Vec<u32>) {
    let mut lens = vec![];
    let mut xy = vec![];
    for pts in list {
        lens.push(pts.len() as u32);
        for p in pts {
            xy.push(p.x);
            xy.push(p.y);
        }
    }
    (lens, xy)
}

fn dump_one(name: &str, w: u32, h: u32, channels: u32, dict_name: &str, raw: Vec<u8>, out_dir: &PathBuf) {
    let image = match channels {
        3 => DynamicImage::ImageRgb8(RgbImage::from_raw(w, h, raw).expect("raw size")),
        4 => DynamicImage::ImageRgba8(RgbaImage::from_raw(w, h, raw).expect("raw size")),
        _ => panic!("channels must be 3 or 4"),
    };
    let detector = Detector { config: DetectorConfig::default(), dictionary: ARDictionary::new_from_named_dict(dict_name) };
    let sample = detector.config.homography_sample_size as u32;
    let window = detector.config.threshold_window;
    let eps = detector.config.contour_simplification_epsilon;
    let mark_size = detector.dictionary.get_mark_size() as u32;

    let mut fx = Fx::new();
    let versions = option_env!("A3_DEP_VERSIONS").unwrap_or("unknown (set A3_DEP_VERSIONS from `cargo tree -e normal --depth 1` when building)");
    fx.u8s("versions", &[versions.len()], versions.as_bytes());

    // ---- the third-party stages, called as Detector::detect calls them ----
    let grey: GrayImage = image.clone().into_luma8();                                            // src/aruco.rs:60
    fx.u8s("grey", &[h as usize, w as usize], grey.as_raw());
    let thresholded = imageproc::contrast::adaptive_threshold(&grey, window);                     // :61
    fx.u8s("thresholded", &[h as usize, w as usize], thresholded.as_raw());
    let contours = imageproc::contours::find_contours::<u32>(&thresholded);                       // :64
    let all_points: Vec<Vec<Point<u32>>> = contours.iter().map(|c| c.points.clone()).collect();
    let (clen, cxy) = points_flat(&all_points);
    fx.u32s("contour_len", &[clen.len()], &clen);
    let border: Vec<u8> = contours.iter().map(|c| if c.border_type == BorderType::Hole { 1u8 } else { 0u8 }).collect();
    fx.u8s("contour_border", &[border.len()], &border);
    fx.u32s("contour_points", &[cxy.len() / 2, 2], &cxy);

    let mut dps: Vec<Vec<Point<u32>>> = vec![];
    let mut hulls: Vec<Vec<Point<u32>>> = vec![];
    for c in contours.iter() {
        let dp = imageproc::geometry::approximate_polygon_dp(&c.points, c.points.len() as f64 * eps, true);   // :133
        let hull = if dp.len() == 4 { imageproc::geometry::convex_hull(dp.clone()) } else { vec![] };           // :143
        dps.push(dp);
        hulls.push(hull);
    }
    let (dlen, dxy) = points_flat(&dps);
    fx.u32s("dp_len", &[dlen.len()], &dlen);
    fx.u32s("dp_points", &[dxy.len() / 2, 2], &dxy);
    let (hlen, hxy) = points_flat(&hulls);
    fx.u32s("hull_len", &[hlen.len()], &hlen);
    fx.u32s("hull_points", &[hxy.len() / 2, 2], &hxy);

    // ---- the crate's own answer ----
    let det = detector.detect(image);
    assert_eq!(det.grey.as_ref().expect("Detection.grey").as_raw(), grey.as_raw());
    let n = det.candidates.len();
    let mut cand = vec![];
    for poly in det.candidates.iter() {
        assert_eq!(poly.len(), 4);
        for p in poly {
            cand.push(p.x);
            cand.push(p.y);
        }
    }
    fx.u32s("candidates", &[n, 4, 2], &cand);

    // ---- per candidate: projection, warp, Otsu, threshold, resize (src/aruco.rs:244-253, 264-273) ----
    let s = sample as usize;
    let (mut src_xy, mut ok, mut homs, mut otsu, mut bins, mut resized) = (vec![], vec![], vec![], vec![], vec![], vec![]);
    let mut equal_detection = 1u8;
    for (i, poly) in det.candidates.iter().enumerate() {
        let hf = sample as f32;
        let projection = Projection::from_control_points(
            [(poly[0].x as f32, poly[0].y as f32), (poly[1].x as f32, poly[1].y as f32), (poly[2].x as f32, poly[2].y as f32), (poly[3].x as f32, poly[3].y as f32)],
            [(0f32, 0f32), (hf, 0f32), (hf, hf), (0f32, hf)],
        );
        let homography = match projection {
            Some(p) => {
                ok.push(1u8);
                let inv = p.invert();
                for y in 0..sample {
                    for x in 0..sample {
                        let (sx, sy) = inv * (x as f32, y as f32);
                        src_xy.push(sx);
                        src_xy.push(sy);
                    }
                }
                let mut out = GrayImage::new(sample, sample);
                warp_into(&grey, &p, Interpolation::Bilinear, [0u8].into(), &mut out);
                out
            }
            None => {
                ok.push(0u8);
                src_xy.extend(std::iter::repeat(f32::NAN).take(s * s * 2));
                GrayImage::new(sample, sample)   // (the crate pushes a 1x1 image here; a full black patch keeps the arrays rectangular)
            }
        };
        if det.homographies[i].dimensions() == (sample, sample) && det.homographies[i].as_raw() != homography.as_raw() {
            equal_detection = 0;
        }
        let level = imageproc::contrast::otsu_level(&homography);                                                        // :264
        let binarized = imageproc::contrast::threshold(&homography, level, imageproc::contrast::ThresholdType::Binary);   // :265
        let reduced = image::imageops::resize(&binarized, mark_size, mark_size, image::imageops::FilterType::Triangle);   // :273
        homs.extend_from_slice(homography.as_raw());
        otsu.push(level);
        bins.extend_from_slice(binarized.as_raw());
        resized.extend_from_slice(reduced.as_raw());
    }
    fx.f32s("warp_src_xy", &[n, s, s, 2], &src_xy);
    fx.u8s("projection_ok", &[n], &ok);
    fx.u8s("homographies", &[n, s, s], &homs);
    fx.u8s("homographies_equal_detection", &[1], &[equal_detection]);
    fx.u8s("otsu", &[n], &otsu);
    fx.u8s("binarized", &[n, s, s], &bins);
    fx.u8s("resized", &[n, mark_size as usize, mark_size as usize], &resized);

    let m = det.markers.len();
    let ids: Vec<u32> = det.markers.iter().map(|k| k.id as u32).collect();
    let codes: Vec<u64> = det.markers.iter().map(|k| k.code).collect();
    let ham: Vec<u8> = det.markers.iter().map(|k| k.hamming_distance).collect();
    let mut corners = vec![];
    for k in det.markers.iter() {
        for c in k.corners.iter() {
            corners.push(c.0);
            corners.push(c.1);
        }
    }
    fx.u32s("marker_id", &[m], &ids);
    fx.u64s("marker_code", &[m], &codes);
    fx.u32s("marker_corners", &[m, 4, 2], &corners);
    fx.u8s("marker_hamming", &[m], &ham);

    let path = out_dir.join(format!("{}.a3fx", name));
    fs::File::create(&path).expect("create").write_all(&fx.buf).expect("write");
    println!("{}: {} contours, {} candidates, {} markers -> {}", name, contours.len(), n, m, path.display());
}

/// Quirk Q4 (src/aruco.rs:255-257): quads no convex hull can deliver, on which `Projection::from_control_points` has no solution, and
/// what the rest of `homography_to_code_permutations` (src/aruco.rs:264-292) makes of the 1 x 1 black image the crate then pushes.
/// The same quads as tests/test_gpu_round6.py (_DEGENERATE), plus two ordinary ones that must solve.  Records (q = quads):
///     q4_quads            u32[q][4][2]
///     q4_projection_ok    u8[q]          from_control_points(quad -> 49 x 49 square) returned Some
///     q4_standin_otsu     u8[1]          otsu_level(&GrayImage::new(1, 1))
///     q4_standin_binary   u8[1]          threshold(.., otsu, Binary) of that image
///     q4_standin_resized  u8[4][10][10]  imageops::resize(.., ms, ms, Triangle) of it for ms = 6, 7, 8, 10 (top-left ms x ms filled)
fn dump_q4(out_dir: &PathBuf) {
    let quads: Vec<[(u32, u32); 4]> = vec![
        [(10, 10), (50, 50), (90, 90), (130, 130)],
        [(300, 300), (300, 300), (340, 300), (340, 340)],
        [(200, 20), (260, 20), (260, 20), (200, 20)],
        [(77, 401), (77, 401), (77, 401), (77, 401)],
        [(500, 100), (560, 100), (620, 100), (560, 160)],
        [(100, 100), (200, 110), (190, 210), (95, 200)],
        [(400, 50), (470, 120), (400, 190), (330, 120)],
    ];
    let mut fx = Fx::new();
    let mut flat = vec![];
    let mut ok = vec![];
    let hf = 49f32;
    for q in quads.iter() {
        for p in q.iter() {
            flat.push(p.0);
            flat.push(p.1);
        }
        let projection = Projection::from_control_points(
            [(q[0].0 as f32, q[0].1 as f32), (q[1].0 as f32, q[1].1 as f32), (q[2].0 as f32, q[2].1 as f32), (q[3].0 as f32, q[3].1 as f32)],
            [(0f32, 0f32), (hf, 0f32), (hf, hf), (0f32, hf)],
        );
        ok.push(if projection.is_some() { 1u8 } else { 0u8 });
    }
    fx.u32s("q4_quads", &[quads.len(), 4, 2], &flat);
    fx.u8s("q4_projection_ok", &[quads.len()], &ok);
    let standin = GrayImage::new(1, 1);                                                                              // src/aruco.rs:256
    let level = imageproc::contrast::otsu_level(&standin);                                                           // :264
    let binarized = imageproc::contrast::threshold(&standin, level, imageproc::contrast::ThresholdType::Binary);     // :265
    fx.u8s("q4_standin_otsu", &[1], &[level]);
    fx.u8s("q4_standin_binary", &[1], binarized.as_raw());
    let mut resized = vec![0u8; 4 * 10 * 10];
    for (k, ms) in [6u32, 7, 8, 10].iter().enumerate() {
        let reduced = image::imageops::resize(&binarized, *ms, *ms, image::imageops::FilterType::Triangle);          // :273
        for y in 0..*ms {
            for x in 0..*ms {
                resized[k * 100 + (y * 10 + x) as usize] = reduced.get_pixel(x, y).0[0];
            }
        }
    }
    fx.u8s("q4_standin_resized", &[4, 10, 10], &resized);
    let path = out_dir.join("q4_degenerate.a3fx");
    fs::File::create(&path).expect("create").write_all(&fx.buf).expect("write");
    println!("q4_degenerate: {:?} -> {}", ok, path.display());
}

#[test]
fn dump_fixtures() {
    let inputs = PathBuf::from(std::env::var("A3_FIXTURE_INPUTS").expect("A3_FIXTURE_INPUTS=<repo>/tests/fixtures/inputs"));
    let out_dir = PathBuf::from(std::env::var("A3_FIXTURE_OUT").expect("A3_FIXTURE_OUT=<repo>/tests/fixtures"));
    let manifest = fs::read_to_string(inputs.join("manifest.txt")).expect("manifest.txt");
    for line in manifest.lines() {
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.len() != 5 {
            continue;
        }
        let (name, w, h, c, dict) = (f[0], f[1].parse::<u32>().unwrap(), f[2].parse::<u32>().unwrap(), f[3].parse::<u32>().unwrap(), f[4]);
        let raw = fs::read(inputs.join(format!("{}.raw", name))).expect("input image");
        assert_eq!(raw.len(), (w * h * c) as usize);
        dump_one(name, w, h, c, dict, raw, &out_dir);
    }
    dump_q4(&out_dir);
}
