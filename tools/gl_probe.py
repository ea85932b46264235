"""Can this box host an OpenGL context that HIP's GL interop accepts?  (the round-4 review's task 5.)

Looks for an EGL implementation the way a host application would (libEGL.so.1 / libEGL.so on the loader path, then any libEGL*
on disk -- pip's kaleido ships SwiftShader's software EGL), asks it for devices (eglQueryDevicesEXT) and for a surfaceless /
default display, creates a context and a buffer object, and -- with that context current on the calling thread -- hands the buffer to
cpm_gl_register_buffer -> cpm_gl_acquire -> cpm_gl_copy_to_buffer -> cpm_gl_release, reading it back with glGetBufferSubData /
glMapBufferRange.  Prints one JSON document: every step's outcome, verbatim error strings included.  Exit code 0 whatever the outcome;
tests/test_gl_gpu.py turns it into a pass (mapped path bit-exact) or a skip that quotes the reason.

usage (GPU box): python tools/gl_probe.py [--no-gpu]      (--no-gpu: stop before the HIP calls; runs anywhere)
"""
import ctypes as C
import ctypes.util
import glob
import json
import os
import sys

EGL_NONE, EGL_OPENGL_ES_API, EGL_OPENGL_API = 0x3038, 0x30A0, 0x30A2
EGL_PLATFORM_DEVICE_EXT, EGL_PLATFORM_SURFACELESS_MESA = 0x313F, 0x31DD
EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES2_BIT, EGL_OPENGL_ES3_BIT, EGL_OPENGL_BIT = 0x3033, 0x0001, 0x3040, 0x0004, 0x0040, 0x0008
EGL_CONTEXT_CLIENT_VERSION, EGL_WIDTH, EGL_HEIGHT, EGL_EXTENSIONS, EGL_VENDOR, EGL_VERSION = 0x3098, 0x3057, 0x3056, 0x3055, 0x3053, 0x3054
GL_ARRAY_BUFFER, GL_PIXEL_UNPACK_BUFFER, GL_DYNAMIC_DRAW, GL_MAP_READ_BIT = 0x8892, 0x88EC, 0x88E8, 0x0001
GL_VENDOR, GL_RENDERER, GL_VERSION = 0x1F00, 0x1F01, 0x1F02


def find_egl():
    found = []
    for name in ("libEGL.so.1", "libEGL.so"):
        try:
            C.CDLL(name)
            found.append(name)
        except OSError:
            pass
    on_disk = []
    for pattern in ("/usr/lib/x86_64-linux-gnu/libEGL*", "/usr/lib64/libEGL*", "/opt/rocm/lib/libEGL*", "/usr/local/lib/python3*/dist-packages/**/libEGL.so",
                    "/opt/conda/lib/python3*/site-packages/**/libEGL.so"):
        on_disk += glob.glob(pattern, recursive=True)
    return found, sorted(set(on_disk))


def probe(no_gpu=False):
    out = {"display_env": os.environ.get("DISPLAY", ""), "dri_nodes": sorted(glob.glob("/dev/dri/*")), "steps": []}

    def step(name, ok, detail=""):
        out["steps"].append({"step": name, "ok": bool(ok), "detail": str(detail)})
        return ok

    loader, disk = find_egl()
    out["egl_on_loader_path"], out["egl_on_disk"] = loader, disk
    out["glx"] = {"libGL": bool(ctypes.util.find_library("GL")), "x_server": bool(os.environ.get("DISPLAY")),
                  "note": "GLX needs an X server; none runs on the box and no Xvfb / Xorg binary is installed"}
    candidates = loader + [d for d in disk if d not in loader]
    if not step("find an EGL library", bool(candidates), candidates or "no libEGL on the loader path or on disk"):
        out["verdict"] = "no EGL implementation on this box: an OpenGL context cannot be created"
        return out
    for path in candidates:
        r = {"library": path}
        out.setdefault("attempts", []).append(r)
        try:
            egl = C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:
            r["load"] = f"failed: {e}"
            continue
        r["load"] = "ok"
        gles = None
        sibling = os.path.join(os.path.dirname(path), "libGLESv2.so") if os.path.isabs(path) else None
        for g in ([sibling] if sibling else []) + ["libGLESv2.so.2", "libGLESv2.so", "libGL.so.1"]:
            try:
                gles = C.CDLL(g, mode=C.RTLD_GLOBAL)
                r["gl_library"] = g
                break
            except OSError:
                continue
        vp = C.c_void_p
        egl.eglGetProcAddress.restype, egl.eglGetProcAddress.argtypes = vp, [C.c_char_p]
        egl.eglGetDisplay.restype, egl.eglGetDisplay.argtypes = vp, [vp]
        egl.eglInitialize.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        egl.eglQueryString.restype, egl.eglQueryString.argtypes = C.c_char_p, [vp, C.c_int]
        egl.eglGetError.restype = C.c_int
        client_ext = egl.eglQueryString(None, EGL_EXTENSIONS)
        r["client_extensions"] = (client_ext or b"").decode()
        # eglQueryDevicesEXT: what devices does this EGL see
        q = egl.eglGetProcAddress(b"eglQueryDevicesEXT")
        if q:
            fn = C.CFUNCTYPE(C.c_uint, C.c_int, C.POINTER(vp), C.POINTER(C.c_int))(q)
            n = C.c_int(0)
            devs = (vp * 16)()
            ok = fn(16, devs, C.byref(n))
            r["eglQueryDevicesEXT"] = f"returned {ok}, {n.value} device(s)" if ok else f"failed, eglGetError 0x{egl.eglGetError():x}"
        else:
            r["eglQueryDevicesEXT"] = "entry point not exported (EGL_EXT_device_enumeration absent)"
        dpy = None
        gpd = egl.eglGetProcAddress(b"eglGetPlatformDisplayEXT")
        if gpd and b"EGL_MESA_platform_surfaceless" in (client_ext or b""):
            fn = C.CFUNCTYPE(vp, C.c_uint, vp, C.POINTER(C.c_int))(gpd)
            dpy = fn(EGL_PLATFORM_SURFACELESS_MESA, None, None)
            r["display"] = "EGL_PLATFORM_SURFACELESS_MESA"
        if not dpy:
            dpy = egl.eglGetDisplay(None)
            r["display"] = "eglGetDisplay(EGL_DEFAULT_DISPLAY)"
        major, minor = C.c_int(0), C.c_int(0)
        if not dpy or not egl.eglInitialize(dpy, C.byref(major), C.byref(minor)):
            r["initialize"] = f"failed, eglGetError 0x{egl.eglGetError():x}"
            continue
        r["initialize"] = f"EGL {major.value}.{minor.value}, vendor {(egl.eglQueryString(dpy, EGL_VENDOR) or b'').decode()}"
        egl.eglBindAPI.argtypes = [C.c_uint]
        api = "OpenGL" if egl.eglBindAPI(EGL_OPENGL_API) else ("OpenGL ES" if egl.eglBindAPI(EGL_OPENGL_ES_API) else None)
        r["api"] = api
        cfg_attr = (C.c_int * 5)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_BIT if api == "OpenGL" else EGL_OPENGL_ES2_BIT, EGL_NONE)
        cfg, ncfg = vp(), C.c_int(0)
        egl.eglChooseConfig.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
        if not egl.eglChooseConfig(dpy, cfg_attr, C.byref(cfg), 1, C.byref(ncfg)) or ncfg.value < 1:
            r["config"] = f"none, eglGetError 0x{egl.eglGetError():x}"
            continue
        egl.eglCreateContext.restype, egl.eglCreateContext.argtypes = vp, [vp, vp, vp, C.POINTER(C.c_int)]
        ctx_attr = (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE) if api != "OpenGL" else (C.c_int * 1)(EGL_NONE)
        glctx = egl.eglCreateContext(dpy, cfg, None, ctx_attr)
        if not glctx and api != "OpenGL":
            ctx_attr = (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 2, EGL_NONE)
            glctx = egl.eglCreateContext(dpy, cfg, None, ctx_attr)
        if not glctx:
            r["context"] = f"failed, eglGetError 0x{egl.eglGetError():x}"
            continue
        egl.eglCreatePbufferSurface.restype, egl.eglCreatePbufferSurface.argtypes = vp, [vp, vp, C.POINTER(C.c_int)]
        pb_attr = (C.c_int * 5)(EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE)
        surf = egl.eglCreatePbufferSurface(dpy, cfg, pb_attr)
        egl.eglMakeCurrent.argtypes = [vp, vp, vp, vp]
        if not egl.eglMakeCurrent(dpy, surf, surf, glctx):
            r["make_current"] = f"failed, eglGetError 0x{egl.eglGetError():x}"
            continue
        r["context"] = "current on the calling thread"

        def glfn(name, res, *args):
            a = egl.eglGetProcAddress(name.encode())
            if not a and gles is not None:
                try:
                    a = C.cast(getattr(gles, name), vp).value
                except AttributeError:
                    a = None
            return C.CFUNCTYPE(res, *args)(a) if a else None
        get_string = glfn("glGetString", C.c_char_p, C.c_uint)
        if get_string:
            r["gl"] = {k: (get_string(v) or b"").decode() for k, v in (("vendor", GL_VENDOR), ("renderer", GL_RENDERER), ("version", GL_VERSION))}
        gen, bind, data = glfn("glGenBuffers", None, C.c_int, C.POINTER(C.c_uint)), glfn("glBindBuffer", None, C.c_uint, C.c_uint), glfn("glBufferData", None, C.c_uint, C.c_ssize_t, vp, C.c_uint)
        if not (gen and bind and data):
            r["buffer"] = "glGenBuffers / glBindBuffer / glBufferData not found"
            continue
        n_texels = 4096
        buf = C.c_uint(0)
        gen(1, C.byref(buf))
        bind(GL_ARRAY_BUFFER, buf)
        data(GL_ARRAY_BUFFER, n_texels * 4, None, GL_DYNAMIC_DRAW)
        r["buffer"] = f"GL buffer {buf.value}, {n_texels * 4} bytes"
        out["context"] = {"library": path, "api": api, "gl": r.get("gl")}
        if no_gpu:
            r["hip"] = "skipped (--no-gpu)"
            break
        # ---- the hand-over itself, through the C-ABI
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import numpy as np
        import torch
        import cpm_amd
        B = cpm_amd.binding
        ctx = B.Context(0)
        r["cpm_gl_available"] = bool(ctx.gl_available())
        try:
            res = ctx.gl_register_buffer(buf.value, False)
        except B.CpmError as e:
            r["cpm_gl_register_buffer"] = f"status {e.status}: {e}"
            out["verdict"] = ("a context could be created (" + path + "), but HIP's GL interop refuses its buffer: " + str(e))
            ctx.close()
            break
        r["cpm_gl_register_buffer"] = "ok"
        vol = torch.arange(n_texels, dtype=torch.float32, device=ctx.device) * 0.25 - 3.0
        try:
            ctx.gl_acquire([res])
            ctx.gl_copy_to_buffer(vol, res)
            ctx.gl_release([res])
            torch.cuda.synchronize()
            r["acquire_copy_release"] = "ok"
        except B.CpmError as e:
            r["acquire_copy_release"] = f"status {e.status}: {e}"
            out["verdict"] = "registered, but map / copy / unmap failed: " + str(e)
            ctx.close()
            break
        back = np.zeros(n_texels, np.float32)
        getsub = glfn("glGetBufferSubData", None, C.c_uint, C.c_ssize_t, C.c_ssize_t, vp)
        mapr, unmap = glfn("glMapBufferRange", vp, C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_uint), glfn("glUnmapBuffer", C.c_ubyte, C.c_uint)
        if getsub:
            getsub(GL_ARRAY_BUFFER, 0, n_texels * 4, back.ctypes.data_as(vp))
        elif mapr and unmap:
            ptr = mapr(GL_ARRAY_BUFFER, 0, n_texels * 4, GL_MAP_READ_BIT)
            if ptr:
                C.memmove(back.ctypes.data, ptr, n_texels * 4)
                unmap(GL_ARRAY_BUFFER)
        same = bool(np.array_equal(back.view(np.uint32), vol.cpu().numpy().view(np.uint32)))
        r["read_back_bit_equal"] = same
        out["verdict"] = "mapped path ran: GL buffer == light volume texels, bit for bit" if same else "mapped path ran but the GL buffer differs"
        out["mapped_path_ok"] = same
        ctx.close()
        break
    if "verdict" not in out:
        out["verdict"] = "no EGL library on this box yields a context (see attempts)" if not out.get("context") else "context created; HIP calls skipped"
    return out


if __name__ == "__main__":
    print(json.dumps(probe("--no-gpu" in sys.argv), indent=1))
