// C# host for libptrace: the parts of the reference's RayTracingMaster.cs that feed the compute
// kernel, with every UnityEngine.ComputeShader / ComputeBuffer call replaced by the matching
// entry point of include/ptrace.h (cited beside each call).
//
// UNVERIFIED SOURCE: no C# toolchain (dotnet, mono) exists where this repository is built and
// tested.  The verified hosts are the C++ mirror (../ray_tracing_master.cpp, same method names)
// and the ctypes harness (../../_ffi.py).  This file shows a maintainer of the reference what the
// binding looks like; struct layouts are the reference's own (RayTracingMaster.cs:93-108).
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using UnityEngine;

public class RayTracingMasterPtrace : MonoBehaviour
{
    [StructLayout(LayoutKind.Sequential)]
    struct MeshObject { public Matrix4x4 localToWorldMatrix; public int indices_offset; public int indices_count; }   // 72 bytes

    [StructLayout(LayoutKind.Sequential)]
    struct Sphere { public Vector3 position; public float radius; public Vector3 albedo; public Vector3 specular; public float smoothness; public Vector3 emission; }   // 56 bytes

    [StructLayout(LayoutKind.Sequential)]
    unsafe struct PtFrameParams
    {
        public fixed float camera_to_world[16];
        public fixed float camera_inverse_projection[16];
        public float seed; public int samples_per_pixel; public int max_bounces; public int depth; public int render_mode;
        public uint sample_index; public int accumulate; public int tile_x0, tile_y0, tile_w, tile_h;
        public fixed float pixel_offset[2]; public fixed float directional_light[4];
        public int accumulation_frames, divisions, counter, left_eye;
    }

    [StructLayout(LayoutKind.Sequential)]
    struct PtCameraHistory { public Matrix4x4 cameraToWorld, cameraInverseProjection, worldToCamera, cameraProjection; }   // 256 bytes

    const string Lib = "ptrace";
    [DllImport(Lib)] static extern int pt_create(int device, out IntPtr ctx);
    [DllImport(Lib)] static extern void pt_destroy(IntPtr ctx);
    [DllImport(Lib)] static extern IntPtr pt_last_error(IntPtr ctx);
    [DllImport(Lib)] static extern int pt_set_spheres(IntPtr ctx, Sphere[] spheres, int count);
    [DllImport(Lib)] static extern int pt_set_meshes(IntPtr ctx, MeshObject[] objects, int nObjects, Vector3[] vertices, int nVertices, int[] indices, int nIndices);
    [DllImport(Lib)] static extern int pt_update_transforms(IntPtr ctx, MeshObject[] objects, int nObjects);
    [DllImport(Lib)] static extern int pt_set_sky(IntPtr ctx, float[] rgba, int width, int height);
    [DllImport(Lib)] static extern int pt_set_camera_history(IntPtr ctx, PtCameraHistory[] entries, int count);
    [DllImport(Lib)] static extern int pt_resize(IntPtr ctx, int width, int height);
    [DllImport(Lib)] static extern int pt_render(IntPtr ctx, ref PtFrameParams p);
    [DllImport(Lib)] static extern int pt_read_converged(IntPtr ctx, float[] rgbaOut);

    public int SphereSeed; public Vector2 SphereRadius = new Vector2(3.0f, 8.0f); public uint SpheresMax = 100; public float SpherePlacementRadius = 100.0f;
    public int samplesPerPixel = 1; public uint sampleFrames = 3; public bool depth = false;
    public int renderMode = 1;                                // every value of RenderMode (:977-990) is implemented
    int divisions = 10, counter = 0;                          // :701-702
    readonly List<PtCameraHistory> _oldCameras = new List<PtCameraHistory>();   // oldIPR / oldCTW / oldWTC / oldPRJ, :450-453

    IntPtr _ctx; Camera _camera; uint _currentSample; int _width, _height; bool _geometryUploaded;
    static readonly List<MeshObject> _meshObjects = new List<MeshObject>();
    static readonly List<Vector3> _vertices = new List<Vector3>();
    static readonly List<int> _indices = new List<int>();
    static bool _meshObjectsNeedRebuilding, _objectListChanged;
    static readonly List<RayTracingObject> _rayTracingObjects = new List<RayTracingObject>();   // :47

    // :296-313, called by RayTracingObject.cs:24-38 exactly as in the reference
    public static int RegisterObject(RayTracingObject obj) { _rayTracingObjects.Add(obj); _meshObjectsNeedRebuilding = _objectListChanged = true; return _rayTracingObjects.Count - 1; }
    public static void UpdateObject(Transform tra) { _meshObjectsNeedRebuilding = true; }
    public static void UnregisterObject(RayTracingObject obj) { _rayTracingObjects.Remove(obj); _meshObjectsNeedRebuilding = _objectListChanged = true; }

    void Check(int rc, string what) { if (rc != 0) throw new InvalidOperationException(what + ": " + Marshal.PtrToStringAnsi(pt_last_error(_ctx))); }

    void OnEnable()
    {
        _camera = GetComponent<Camera>();
        Check(pt_create(0, out _ctx), "pt_create");
        _currentSample = 0;                                   // RayTracingMaster.cs:145
        SetUpScene();                                         // :146
    }

    void OnDisable() { if (_ctx != IntPtr.Zero) { pt_destroy(_ctx); _ctx = IntPtr.Zero; } }   // :154-161

    void SetUpScene()                                         // :315-369, unchanged up to the upload
    {
        UnityEngine.Random.InitState(SphereSeed);
        var spheres = new List<Sphere>();
        for (int i = 0; i < SpheresMax; i++)
        {
            var s = new Sphere();
            s.radius = SphereRadius.x + UnityEngine.Random.value * (SphereRadius.y - SphereRadius.x);
            Vector2 p = UnityEngine.Random.insideUnitCircle * SpherePlacementRadius;
            s.position = new Vector3(p.x, s.radius, p.y);
            bool skip = false;
            foreach (var o in spheres) { float m = s.radius + o.radius; if (Vector3.SqrMagnitude(s.position - o.position) < m * m) { skip = true; break; } }
            if (skip) continue;
            Color c = UnityEngine.Random.ColorHSV(); float chance = UnityEngine.Random.value;
            if (chance < 0.8f) { bool metal = chance < 0.4f; s.albedo = metal ? Vector3.zero : new Vector3(c.r, c.g, c.b); s.specular = metal ? new Vector3(c.r, c.g, c.b) : new Vector3(0.04f, 0.04f, 0.04f); s.smoothness = UnityEngine.Random.value; }
            else { Color e = UnityEngine.Random.ColorHSV(0, 1, 0, 1, 3.0f, 8.0f); s.emission = new Vector3(e.r, e.g, e.b); }
            spheres.Add(s);
        }
        Check(pt_set_spheres(_ctx, spheres.ToArray(), spheres.Count), "pt_set_spheres");   // replaces :362-368 + :536
    }

    void RebuildMeshObjectBuffers()                            // :371-413
    {
        if (!_meshObjectsNeedRebuilding) return;
        _meshObjectsNeedRebuilding = false;
        _meshObjects.Clear();
        if (_objectListChanged) { _vertices.Clear(); _indices.Clear(); }
        foreach (RayTracingObject obj in _rayTracingObjects)   // :384-405
        {
            Mesh mesh = obj.GetComponent<MeshFilter>().sharedMesh;
            int firstVertex = _vertices.Count, firstIndex = _indices.Count;
            int[] indices = mesh.GetIndices(0);
            if (_objectListChanged) { _vertices.AddRange(mesh.vertices); foreach (int i in indices) _indices.Add(i + firstVertex); }
            else firstIndex = _meshObjects.Count == 0 ? 0 : _meshObjects[_meshObjects.Count - 1].indices_offset + _meshObjects[_meshObjects.Count - 1].indices_count;
            _meshObjects.Add(new MeshObject { localToWorldMatrix = obj.transform.localToWorldMatrix, indices_offset = firstIndex, indices_count = indices.Length });
        }
        if (_objectListChanged) { _geometryUploaded = false; _objectListChanged = false; }
        if (!_geometryUploaded)
        {
            Check(pt_set_meshes(_ctx, _meshObjects.ToArray(), _meshObjects.Count, _vertices.ToArray(), _vertices.Count, _indices.ToArray(), _indices.Count), "pt_set_meshes");   // :410-412 + :537-539
            _geometryUploaded = true;
        }
        else Check(pt_update_transforms(_ctx, _meshObjects.ToArray(), _meshObjects.Count), "pt_update_transforms");
    }

    unsafe void Render()                                       // :704-747
    {
        RebuildMeshObjectBuffers();                            // :714
        int w = Screen.width, h = Screen.height;
        if (w != _width || h != _height) { Check(pt_resize(_ctx, w, h), "pt_resize"); _width = w; _height = h; }   // :542-583
        var p = new PtFrameParams();
        // :459-467 / :489-490 - a stereo camera renders one eye per instance, with that eye's own view and projection
        bool stereo = _camera.stereoEnabled, left = _camera.stereoTargetEye == StereoTargetEyeMask.Left;
        Camera.StereoscopicEye which = left ? Camera.StereoscopicEye.Left : Camera.StereoscopicEye.Right;
        Matrix4x4 w2c = stereo ? _camera.GetStereoViewMatrix(which) : _camera.worldToCameraMatrix;
        Matrix4x4 prj = stereo ? _camera.GetStereoProjectionMatrix(which) : _camera.projectionMatrix;
        Matrix4x4 c2w = stereo ? w2c.inverse : _camera.cameraToWorldMatrix, ip = prj.inverse;
        for (int i = 0; i < 16; i++) { p.camera_to_world[i] = c2w[i]; p.camera_inverse_projection[i] = ip[i]; }  // Matrix4x4[i] is column-major
        p.seed = UnityEngine.Random.value;                     // :530
        p.samples_per_pixel = renderMode < 10 ? samplesPerPixel : 1;   // :531
        p.max_bounces = 4;                                     // the shader's loop bound, RayTracingShader.compute:867
        p.depth = depth ? 1 : 0;                               // :532
        p.render_mode = renderMode;                            // :735
        p.accumulation_frames = (int)sampleFrames;             // :738
        p.divisions = divisions; p.counter = counter++ % divisions;   // :736-737
        p.left_eye = stereo && left ? 1 : 0;                   // :460-462 (uploaded by the stereo branch only)
        p.accumulate = (depth || renderMode != 1) ? 0 : 3;     // AddMat.shader blend; in Reproj mode _converged is the history
        Check(pt_set_camera_history(_ctx, _oldCameras.ToArray(), _oldCameras.Count), "pt_set_camera_history");   // :468-471 / :492-495
        // when the lists restart: the stereo branch keeps them only in modes 5 / 6 and only across the frame that starts
        // a cycle (:473); the mono branch restarts them at the start of a cycle in modes 5 .. 8, every frame otherwise (:513)
        bool clear = stereo ? ((renderMode != 5 && renderMode != 6) || p.counter != 0)
                            : (renderMode < 5 || renderMode > 8 || p.counter == 0);
        if (clear) _oldCameras.Clear();
        if (renderMode < 11 && renderMode != 1)                                                                  // :480-485 / :520-526
            _oldCameras.Add(new PtCameraHistory { cameraToWorld = c2w, cameraInverseProjection = ip, worldToCamera = w2c, cameraProjection = prj });
        p.sample_index = depth ? 0u : _currentSample;          // :805
        Check(pt_render(_ctx, ref p), "pt_render");            // :745-747
        if (_currentSample < sampleFrames) _currentSample++; else _currentSample = sampleFrames;   // :885-888
    }
}
