//! The public API of JakubValtar/basisu_rs (src/lib.rs:20-79 of the reference) over libbasisu_hip.so.
//! NOT COMPILED HERE: the build image has no Rust toolchain.  See ../README.md.
//!
//! Same names, signatures and error strings as the reference:
//!   read_to_rgba / read_to_etc1 / read_to_etc2 / read_to_uastc / read_to_astc / read_to_bc7   (basis.rs:8-260)
//!   unpack_uastc_block_to_rgba, transcode_uastc_block_to_{astc,bc7,etc1,etc2}                   (lib.rs:29-53)
//!   Image<T>, Header                                                                            (lib.rs:63-68, basis.rs:417-473)
//! plus `transcode_array_sharded`, the multi-device entry the reference has no counterpart for.
pub mod ffi;

use std::ffi::CStr;
use std::os::raw::c_int;
use std::sync::OnceLock;

pub type Error = String;
pub type Result<T> = core::result::Result<T, Error>;

pub const UASTC_BLOCK_SIZE: usize = 16;
pub const ASTC_BLOCK_SIZE: usize = 16;
pub const BC7_BLOCK_SIZE: usize = 16;
pub const ETC1_BLOCK_SIZE: usize = 8;
pub const ETC2_BLOCK_SIZE: usize = 16;

/// lib.rs:63-68
pub struct Image<T> {
    pub w: u32,
    pub h: u32,
    pub stride: u32,
    pub data: Vec<T>,
}

/// basis::Header (basis.rs:417-454): the C mirror has the same 26 public fields
pub type Header = ffi::bu_basis_header;

impl ffi::bu_basis_header {
    pub const FILE_SIZE: usize = 77;
    /// basis.rs:463-465 (HeaderFlags::HasAlphaSlices = 4)
    pub fn has_alpha(&self) -> bool {
        self.flags & 4 != 0
    }
    /// basis.rs:467-469 (HeaderFlags::YFlipped = 2)
    pub fn has_y_flipped(&self) -> bool {
        self.flags & 2 != 0
    }
}

struct Ctx(*mut ffi::bu_context);
// the C side serialises host-pointer calls on one context with a mutex and keeps no thread-local state
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

fn status_string(st: c_int) -> String {
    unsafe { CStr::from_ptr(ffi::bu_status_string(st)) }.to_string_lossy().into_owned()
}

fn context() -> Result<*mut ffi::bu_context> {
    static CTX: OnceLock<core::result::Result<Ctx, String>> = OnceLock::new();
    let r = CTX.get_or_init(|| {
        let device = std::env::var("BASISU_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut p = core::ptr::null_mut();
        let st = unsafe { ffi::bu_context_create(device, &mut p) };
        if st == ffi::BU_OK { Ok(Ctx(p)) } else { Err(status_string(st)) }
    });
    match r {
        Ok(c) => Ok(c.0),
        Err(e) => Err(e.clone()),
    }
}

fn check(ctx: *mut ffi::bu_context, st: c_int) -> Result<()> {
    if st == ffi::BU_OK {
        return Ok(());
    }
    let mut msg = status_string(st); // the reference's text for its own errors (uastc.rs:56,336,364; basis.rs:12,309-404)
    if st == 8 {
        // BU_ERR_HIP: detail from the runtime
        msg.push_str(": ");
        msg.push_str(&unsafe { CStr::from_ptr(ffi::bu_last_error(ctx)) }.to_string_lossy());
    }
    Err(msg)
}

fn read_to(target: c_int, buf: &[u8]) -> Result<(Header, Vec<Image<u8>>)> {
    let ctx = context()?;
    let (mut n_images, mut out_bytes) = (0usize, 0usize);
    check(ctx, unsafe { ffi::bu_read_query(target, buf.as_ptr(), buf.len(), &mut n_images, &mut out_bytes) })?;
    let mut out = vec![0u8; out_bytes.max(1)];
    let mut images = vec![ffi::bu_image::default(); n_images.max(1)];
    let mut header = Header::default();
    let st = unsafe {
        ffi::bu_read_to(ctx, target, buf.as_ptr(), buf.len(), &mut header, images.as_mut_ptr(), n_images, &mut n_images, out.as_mut_ptr(), out.len())
    };
    check(ctx, st)?; // an Err drops every image, like the `?` inside the reference's slice loops
    let imgs = images[..n_images]
        .iter()
        .map(|im| Image { w: im.w, h: im.h, stride: im.stride, data: out[im.offset as usize..(im.offset + im.size) as usize].to_vec() })
        .collect();
    Ok((header, imgs))
}

/// basis.rs:8-90
pub fn read_to_rgba(buf: &[u8]) -> Result<(Header, Vec<Image<u8>>)> {
    read_to(ffi::BU_READ_RGBA, buf)
}
/// basis.rs:92-143
pub fn read_to_etc1(buf: &[u8]) -> Result<Vec<Image<u8>>> {
    read_to(ffi::BU_READ_ETC1, buf).map(|r| r.1)
}
/// basis.rs:145-173
pub fn read_to_etc2(buf: &[u8]) -> Result<Vec<Image<u8>>> {
    read_to(ffi::BU_READ_ETC2, buf).map(|r| r.1)
}
/// basis.rs:175-202
pub fn read_to_uastc(buf: &[u8]) -> Result<Vec<Image<u8>>> {
    read_to(ffi::BU_READ_UASTC, buf).map(|r| r.1)
}
/// basis.rs:204-231
pub fn read_to_astc(buf: &[u8]) -> Result<Vec<Image<u8>>> {
    read_to(ffi::BU_READ_ASTC, buf).map(|r| r.1)
}
/// basis.rs:233-260
pub fn read_to_bc7(buf: &[u8]) -> Result<Vec<Image<u8>>> {
    read_to(ffi::BU_READ_BC7, buf).map(|r| r.1)
}

/// lib.rs:29-31
pub fn unpack_uastc_block_to_rgba(data: [u8; UASTC_BLOCK_SIZE]) -> Result<[u32; 16]> {
    let ctx = context()?;
    let mut out = [0u32; 16];
    check(ctx, unsafe { ffi::bu_unpack_uastc_block_to_rgba(ctx, data.as_ptr(), out.as_mut_ptr()) })?;
    Ok(out)
}
/// lib.rs:33-37
pub fn transcode_uastc_block_to_astc(data: [u8; UASTC_BLOCK_SIZE]) -> Result<[u8; ASTC_BLOCK_SIZE]> {
    let ctx = context()?;
    let mut out = [0u8; ASTC_BLOCK_SIZE];
    check(ctx, unsafe { ffi::bu_transcode_uastc_block_to_astc(ctx, data.as_ptr(), out.as_mut_ptr()) })?;
    Ok(out)
}
/// lib.rs:39-41
pub fn transcode_uastc_block_to_bc7(data: [u8; UASTC_BLOCK_SIZE]) -> Result<[u8; BC7_BLOCK_SIZE]> {
    let ctx = context()?;
    let mut out = [0u8; BC7_BLOCK_SIZE];
    check(ctx, unsafe { ffi::bu_transcode_uastc_block_to_bc7(ctx, data.as_ptr(), out.as_mut_ptr()) })?;
    Ok(out)
}
/// lib.rs:43-47
pub fn transcode_uastc_block_to_etc1(data: [u8; UASTC_BLOCK_SIZE]) -> Result<[u8; ETC1_BLOCK_SIZE]> {
    let ctx = context()?;
    let mut out = [0u8; ETC1_BLOCK_SIZE];
    check(ctx, unsafe { ffi::bu_transcode_uastc_block_to_etc1(ctx, data.as_ptr(), out.as_mut_ptr()) })?;
    Ok(out)
}
/// lib.rs:49-53
pub fn transcode_uastc_block_to_etc2(data: [u8; UASTC_BLOCK_SIZE]) -> Result<[u8; ETC2_BLOCK_SIZE]> {
    let ctx = context()?;
    let mut out = [0u8; ETC2_BLOCK_SIZE];
    check(ctx, unsafe { ffi::bu_transcode_uastc_block_to_etc2(ctx, data.as_ptr(), out.as_mut_ptr()) })?;
    Ok(out)
}

/// uastc::TargetTextureFormat (uastc.rs:41-47)
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum TargetTextureFormat {
    Astc = 0,
    Bc7 = 1,
    Etc1 = 2,
    Etc2 = 3,
}

/// uastc::Decoder::transcode (uastc.rs:112-121): one slice, host memory in and out
pub fn transcode_slice(format: TargetTextureFormat, data: &[u8]) -> Result<Vec<u8>> {
    let ctx = context()?;
    let bb = unsafe { ffi::bu_target_block_bytes(format as c_int) };
    let mut out = vec![0u8; data.len() / UASTC_BLOCK_SIZE * bb];
    let mut bad = 0u64;
    check(ctx, unsafe { ffi::bu_uastc_transcode(ctx, format as c_int, data.as_ptr(), data.len(), out.as_mut_ptr(), out.len(), &mut bad) })?;
    Ok(out)
}

/// Texture array sharded over the devices `devices` of one node (no counterpart in the reference, which walks the slices
/// one by one: basis.rs:246-257): slice range g*n/P..(g+1)*n/P goes to device g, every device ends up holding the whole
/// transcoded array (`gather = true`: peer pulls over xGMI), and the array is returned from device 0.
pub fn transcode_array_sharded(format: TargetTextureFormat, slices: &[u8], n_slices: usize, devices: &[i32], gather: bool) -> Result<Vec<u8>> {
    use core::ffi::c_void;
    if devices.is_empty() || n_slices == 0 || slices.len() % (UASTC_BLOCK_SIZE * n_slices) != 0 {
        return Err("transcode_array_sharded: need at least one device, one slice and whole slices of whole blocks".to_string());
    }
    let bps = slices.len() / UASTC_BLOCK_SIZE / n_slices.max(1);
    let bb = unsafe { ffi::bu_target_block_bytes(format as c_int) };
    let n = devices.len();
    let mut ctxs = Vec::with_capacity(n);
    for &d in devices {
        let mut p = core::ptr::null_mut();
        let st = unsafe { ffi::bu_context_create(d, &mut p) };
        if st != ffi::BU_OK {
            for &c in &ctxs {
                unsafe { ffi::bu_context_destroy(c) };
            }
            return Err(status_string(st));
        }
        ctxs.push(p);
    }
    let total_out = n_slices * bps * bb;
    let (mut d_in, mut d_full): (Vec<*mut c_void>, Vec<*mut c_void>) = (vec![core::ptr::null_mut(); n], vec![core::ptr::null_mut(); n]);
    let result = (|| -> Result<Vec<u8>> {
        for g in 0..n {
            let (lo, hi) = (n_slices * g / n, n_slices * (g + 1) / n);
            let bytes = (hi - lo) * bps * UASTC_BLOCK_SIZE;
            check(ctxs[g], unsafe { ffi::bu_device_alloc(ctxs[g], bytes.max(16), &mut d_in[g]) })?;
            check(ctxs[g], unsafe { ffi::bu_device_alloc(ctxs[g], total_out.max(16), &mut d_full[g]) })?;
            let src = &slices[lo * bps * UASTC_BLOCK_SIZE..hi * bps * UASTC_BLOCK_SIZE];
            check(ctxs[g], unsafe { ffi::bu_memcpy(ctxs[g], d_in[g], src.as_ptr() as *const c_void, src.len(), 1) })?;
        }
        let mut bad = 0u64;
        let ins: Vec<*const c_void> = d_in.iter().map(|p| *p as *const c_void).collect();
        let st = unsafe {
            ffi::bu_array_transcode_sharded(ctxs.as_ptr(), n as c_int, format as c_int, ins.as_ptr(), n_slices, bps, d_full.as_ptr(), gather as c_int, &mut bad)
        };
        check(ctxs[0], st)?;
        let mut out = vec![0u8; total_out];
        if gather {
            check(ctxs[0], unsafe { ffi::bu_memcpy(ctxs[0], out.as_mut_ptr() as *mut c_void, d_full[0], total_out, 0) })?;
        } else {
            for g in 0..n {
                let (lo, hi) = (n_slices * g / n * bps * bb, n_slices * (g + 1) / n * bps * bb);
                let src = unsafe { (d_full[g] as *const u8).add(lo) } as *const c_void;
                check(ctxs[g], unsafe { ffi::bu_memcpy(ctxs[g], out[lo..hi].as_mut_ptr() as *mut c_void, src, hi - lo, 0) })?;
            }
        }
        Ok(out)
    })();
    for g in 0..n {
        unsafe {
            ffi::bu_device_free(ctxs[g], d_in[g]);
            ffi::bu_device_free(ctxs[g], d_full[g]);
            ffi::bu_context_destroy(ctxs[g]);
        }
    }
    result
}
