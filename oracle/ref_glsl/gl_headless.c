/* TEST INFRASTRUCTURE — not product code.
 *
 * X-less OpenGL 3.3+ core context on Mesa's llvmpipe, for executing the reference's GLSL shaders
 * (/root/reference/Core/src/Shaders/, read at run time by ref_glsl.py, never copied into this repository)
 * in the build container. The reference creates its context through Pangolin + GLX (GUI/src/GUI.h); neither
 * Pangolin nor an X server exists here, and none is needed: swrast_dri.so exports the DRI_SWRast / DRI_Core
 * driver extensions, which is what libGL's own software path binds. Nothing here stands in for a reference
 * header or library — the harness only owns a GL context; the arithmetic is the reference's shader text compiled
 * by Mesa's GLSL compiler.
 *
 * Built by oracle/ref_glsl/Makefile into oracle/_ref/libgl_headless.so. Only tests/golden/make_ref_glsl.py
 * (fixture generation, this container only) loads it.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <GL/gl.h>
#include <GL/internal/dri_interface.h>

static void *g_drv, *g_glapi;
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static int g_w = 16, g_h = 16;

static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = g_w; *h = g_h; }
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void ld_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ld_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
    .putImage2 = ld_put_image2,
    .getImage2 = ld_get_image2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

/* returns 0 on success; the GL_VERSION string can then be read through glh_proc("glGetString") */
int glh_init(const char *driver_path, int compat)
{
    if (g_ctx) return 0;
    /* the dispatch library first, global, so that the driver resolves _glapi_* against it */
    g_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!g_glapi) { fprintf(stderr, "glh: %s\n", dlerror()); return 1; }
    g_drv = dlopen(driver_path ? driver_path : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!g_drv) { fprintf(stderr, "glh: %s\n", dlerror()); return 2; }
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(g_drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) { fprintf(stderr, "glh: no __driDriverGetExtensions_swrast\n"); return 3; }
    const __DRIextension **exts = get_exts();
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) { fprintf(stderr, "glh: driver extensions missing\n"); return 4; }
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) { fprintf(stderr, "glh: createNewScreen2 failed\n"); return 5; }
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, compat ? __DRI_API_OPENGL : __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { fprintf(stderr, "glh: createContextAttribs failed (%u)\n", err); return 6; }
    g_draw = g_swrast->createNewDrawable(g_screen, configs[0], NULL);
    if (!g_draw) { fprintf(stderr, "glh: createNewDrawable failed\n"); return 7; }
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { fprintf(stderr, "glh: bindContext failed\n"); return 8; }
    return 0;
}

void *glh_proc(const char *name)
{
    static void *(*gpa)(const char *);
    if (!gpa) gpa = (void *(*)(const char *))dlsym(g_glapi, "_glapi_get_proc_address");
    return gpa ? gpa(name) : NULL;
}
