// scene_loader.cpp — see scene_loader.h.
#include "scene_loader.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "mini_xml.h"

namespace tk {

namespace {

const float kPi = float(4.0 * std::atan(1.0));  // src/TkMath.cpp:7

std::string dirName(const std::string& p) {
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
bool fileExists(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return f.good();
}
std::string readFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

[[noreturn]] void failAt(const XmlNode* n, const std::string& what) {
    throw std::runtime_error("ERROR, line " + std::to_string(n->line) + "; in " + n->name + ", " + what);
}

// get_unique_child, src/TkLoader.cpp:64-86
const XmlNode* uniqueChild(const XmlNode* parent, bool required, const char* name) {
    auto v = parent->all(name);
    if (v.empty()) {
        if (required) failAt(parent, std::string("no '") + name + "' defined.");
        return nullptr;
    }
    if (v.size() > 1) failAt(v[1], std::string("'") + name + "' multiply defined.");
    return v[0];
}
// parse_attrib_real, src/TkLoader.cpp:88-105
bool attribReal(const XmlNode* e, bool required, const char* name, float* val) {
    const std::string* a = e->attr(name);
    if (!a) {
        if (required) failAt(e, std::string("missing '") + name + "'.");
        return false;
    }
    char* end = nullptr;
    double d = std::strtod(a->c_str(), &end);
    if (end == a->c_str()) failAt(e, std::string("error parsing '") + name + "'.");
    *val = float(d);
    return true;
}
// parse_attrib_int, src/TkLoader.cpp:107-120
bool attribInt(const XmlNode* e, bool required, const char* name, int32_t* val) {
    const std::string* a = e->attr(name);
    if (!a) {
        if (required) failAt(e, std::string("missing '") + name + "'.");
        return false;
    }
    char* end = nullptr;
    long v = std::strtol(a->c_str(), &end, 10);
    if (end == a->c_str()) failAt(e, std::string("error parsing '") + name + "'.");
    *val = int32_t(v);
    return true;
}
bool attribString(const XmlNode* e, bool required, const char* name, std::string* val) {
    const std::string* a = e->attr(name);
    if (!a) {
        if (required) failAt(e, std::string("missing '") + name + "'.");
        return false;
    }
    *val = *a;
    return true;
}
// parse_elem<Real> / <s32> / <Spectrum> / <Vector3f> on a unique child
// (src/TkLoader.cpp:150-201)
void childReal(const XmlNode* p, bool required, const char* name, float* v) {
    if (const XmlNode* c = uniqueChild(p, required, name)) attribReal(c, true, "v", v);
}
void childInt(const XmlNode* p, bool required, const char* name, int32_t* v) {
    if (const XmlNode* c = uniqueChild(p, required, name)) attribInt(c, true, "i", v);
}
void childColor(const XmlNode* p, bool required, const char* name, float rgb[3]) {
    if (const XmlNode* c = uniqueChild(p, required, name)) {
        attribReal(c, true, "r", &rgb[0]);
        attribReal(c, true, "g", &rgb[1]);
        attribReal(c, true, "b", &rgb[2]);
    }
}
bool childVec3(const XmlNode* p, bool required, const char* name, float v[3]) {
    if (const XmlNode* c = uniqueChild(p, required, name)) {
        attribReal(c, true, "x", &v[0]);
        attribReal(c, true, "y", &v[1]);
        attribReal(c, true, "z", &v[2]);
        return true;
    }
    return false;
}

// parse_material, src/TkLoader.cpp:281-318
TkMaterial parseMaterial(const XmlNode* e, std::string* name) {
    std::string type;
    attribString(e, true, "name", name);
    attribString(e, true, "type", &type);
    TkMaterial m{};
    m.R[0] = m.R[1] = m.R[2] = 1.f;
    m.T[0] = m.T[1] = m.T[2] = 1.f;
    m.etaA = 1.f;
    m.etaB = 1.f;
    if (type == "glass") {
        m.type = TK_GLASS;
        childReal(e, false, "refractive_index", &m.etaB);
        childColor(e, false, "R", m.R);
        childColor(e, false, "T", m.T);
    } else if (type == "mirror") {
        m.type = TK_MIRROR;
        childColor(e, false, "R", m.R);
    } else if (type == "diffuse") {
        m.type = TK_DIFFUSE;
        childColor(e, false, "R", m.R);
    } else {
        // plastic is only used by the bidirectional scene, which this path does not render
        failAt(e, "material type '" + type + "' is not supported by the SPPM device path.");
    }
    return m;
}

struct MeshEntry {
    std::vector<TkTriangle> tris;
};

}  // namespace

// Quaternion::setByEuler + getMatrix + Matrix4::makeTransform
// (src/Quaternion.cpp:47-108, src/Matrix4.cpp:105-116)
Mat4 makeTransform(const float translate[3], const float scale[3], const float euler[3]) {
    float pitch = float(euler[0] * 0.5), yaw = float(euler[1] * 0.5), roll = float(euler[2] * 0.5);
    float sp = std::sin(pitch), cp = std::cos(pitch);
    float sy = std::sin(yaw), cy = std::cos(yaw);
    float sr = std::sin(roll), cr = std::cos(roll);
    float crcy = cr * cy, srsy = sr * sy, crsy = cr * sy, srcy = sr * cy;
    float w = crcy * cp + srsy * sp;
    float x = crcy * sp - srsy * cp;
    float y = crsy * cp + srcy * sp;
    float z = srcy * cp - crsy * sp;
    float l = std::sqrt(w * w + x * x + y * y + z * z);
    float invL = 1.0f / l;
    w *= invL; x *= invL; y *= invL; z *= invL;
    float fTx = x + x, fTy = y + y, fTz = z + z;
    float fTwx = fTx * w, fTwy = fTy * w, fTwz = fTz * w;
    float fTxx = fTx * x, fTxy = fTy * x, fTxz = fTz * x;
    float fTyy = fTy * y, fTyz = fTz * y, fTzz = fTz * z;
    float rot[9] = {1.0f - (fTyy + fTzz), fTxy - fTwz, fTxz + fTwy,
                    fTxy + fTwz, 1.0f - (fTxx + fTzz), fTyz - fTwx,
                    fTxz - fTwy, fTyz + fTwx, 1.0f - (fTxx + fTyy)};
    Mat4 r;
    for (int row = 0; row < 3; ++row) {
        r.m[row * 4 + 0] = scale[0] * rot[row * 3 + 0];
        r.m[row * 4 + 1] = scale[1] * rot[row * 3 + 1];
        r.m[row * 4 + 2] = scale[2] * rot[row * 3 + 2];
        r.m[row * 4 + 3] = translate[row];
    }
    r.m[12] = r.m[13] = r.m[14] = 0;
    r.m[15] = 1;
    return r;
}

// Matrix4::concatenatePos, src/Matrix4.cpp:118-128
void transformPos(const Mat4& mt, const float p[3], float out[3]) {
    const float* m = mt.m;
    float xp = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
    float yp = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
    float zp = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
    float wp = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
    if (wp == 1) {
        out[0] = xp; out[1] = yp; out[2] = zp;
    } else {
        out[0] = xp / wp; out[1] = yp / wp; out[2] = zp / wp;
    }
}

namespace {

void pushTriangle(const Mat4& xf, const float a[3], const float b[3], const float c[3], std::vector<TkTriangle>* out) {
    TkTriangle t;
    transformPos(xf, a, t.v0);
    transformPos(xf, b, t.v1);
    transformPos(xf, c, t.v2);
    out->push_back(t);
}

// Wavefront OBJ: "v x y z" and "f i[/t[/n]] ..." records; everything else is
// ignored.  The reference consumes the face corners three at a time
// (src/Triangle.cpp:117-137), which is one triangle per "f" line for
// triangulated files; faces with more corners are fanned here.
void loadObj(const std::string& path, const Mat4& xf, std::vector<TkTriangle>* out) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<float> verts;
    std::string line;
    std::vector<long> corner;
    while (std::getline(f, line)) {
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char* end = nullptr;
            const char* p = s + 1;
            for (int k = 0; k < 3; ++k) {
                float v = std::strtof(p, &end);
                if (end == p) throw std::runtime_error(path + ": malformed vertex: " + line);
                verts.push_back(v);
                p = end;
            }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            corner.clear();
            const char* p = s + 1;
            for (;;) {
                while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
                if (!*p) break;
                char* end = nullptr;
                long idx = std::strtol(p, &end, 10);
                if (end == p) throw std::runtime_error(path + ": malformed face: " + line);
                long nv = long(verts.size() / 3);
                if (idx < 0) idx = nv + idx; else idx = idx - 1;
                if (idx < 0 || idx >= nv) throw std::runtime_error(path + ": face index out of range: " + line);
                corner.push_back(idx);
                p = end;
                while (*p && *p != ' ' && *p != '\t' && *p != '\r') ++p;  // skip /t/n
            }
            if (corner.size() < 3) throw std::runtime_error(path + ": face with fewer than 3 corners");
            for (size_t k = 1; k + 1 < corner.size(); ++k)
                pushTriangle(xf, &verts[3 * corner[0]], &verts[3 * corner[k]], &verts[3 * corner[k + 1]], out);
        }
    }
}

size_t plyTypeSize(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    throw std::runtime_error("ply: unknown type " + t);
}
double plyReadBinary(const unsigned char* p, const std::string& t, bool bigEndian) {
    unsigned char b[8];
    size_t n = plyTypeSize(t);
    for (size_t k = 0; k < n; ++k) b[k] = bigEndian ? p[n - 1 - k] : p[k];
    if (t == "char" || t == "int8") { int8_t v; memcpy(&v, b, 1); return v; }
    if (t == "uchar" || t == "uint8") { uint8_t v; memcpy(&v, b, 1); return v; }
    if (t == "short" || t == "int16") { int16_t v; memcpy(&v, b, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, b, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; memcpy(&v, b, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; memcpy(&v, b, 4); return v; }
    if (t == "float" || t == "float32") { float v; memcpy(&v, b, 4); return v; }
    double v; memcpy(&v, b, 8); return v;
}

// PLY (ascii, binary_little_endian, binary_big_endian): vertex x,y,z and a
// face index list; triangles and quads (split 0-1-2, 0-2-3) as createPLYMesh
// does (src/PLY_Loader.cpp), larger polygons are fanned.
void loadPly(const std::string& path, const Mat4& xf, std::vector<TkTriangle>* out) {
    std::string data = readFile(path);
    size_t pos = 0;
    auto nextLine = [&](std::string* l) {
        if (pos >= data.size()) return false;
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        *l = data.substr(pos, e - pos);
        if (!l->empty() && l->back() == '\r') l->pop_back();
        pos = e + 1;
        return true;
    };
    struct Prop { std::string name, type, countType; bool isList = false; };
    struct Elem { std::string name; long count = 0; std::vector<Prop> props; };
    std::vector<Elem> elems;
    std::string line, format;
    if (!nextLine(&line) || line != "ply") throw std::runtime_error(path + ": not a PLY file");
    while (nextLine(&line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "format") ss >> format;
        else if (tok == "element") { Elem e; ss >> e.name >> e.count; elems.push_back(e); }
        else if (tok == "property") {
            if (elems.empty()) throw std::runtime_error(path + ": property before element");
            Prop p; ss >> p.type;
            if (p.type == "list") { p.isList = true; ss >> p.countType >> p.type; }
            ss >> p.name;
            elems.back().props.push_back(p);
        } else if (tok == "end_header") break;
    }
    bool ascii = format == "ascii", big = format == "binary_big_endian";
    if (!ascii && !big && format != "binary_little_endian") throw std::runtime_error(path + ": unknown PLY format " + format);
    std::vector<float> verts;
    std::istringstream as;
    if (ascii) as.str(data.substr(pos));
    const unsigned char* bp = reinterpret_cast<const unsigned char*>(data.data());
    auto readScalar = [&](const std::string& type) -> double {
        if (ascii) { double v; if (!(as >> v)) throw std::runtime_error(path + ": truncated PLY body"); return v; }
        size_t n = plyTypeSize(type);
        if (pos + n > data.size()) throw std::runtime_error(path + ": truncated PLY body");
        double v = plyReadBinary(bp + pos, type, big);
        pos += n;
        return v;
    };
    for (const Elem& e : elems) {
        for (long i = 0; i < e.count; ++i) {
            float xyz[3] = {0, 0, 0};
            std::vector<long> idx;
            for (const Prop& p : e.props) {
                if (p.isList) {
                    long n = long(readScalar(p.countType));
                    std::vector<long> tmp;
                    for (long k = 0; k < n; ++k) tmp.push_back(long(readScalar(p.type)));
                    if (e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index")) idx = tmp;
                } else {
                    double v = readScalar(p.type);
                    if (e.name == "vertex") {
                        if (p.name == "x") xyz[0] = float(v);
                        else if (p.name == "y") xyz[1] = float(v);
                        else if (p.name == "z") xyz[2] = float(v);
                    }
                }
            }
            if (e.name == "vertex") { verts.push_back(xyz[0]); verts.push_back(xyz[1]); verts.push_back(xyz[2]); }
            else if (e.name == "face" && idx.size() >= 3) {
                long nv = long(verts.size() / 3);
                for (long v : idx) if (v < 0 || v >= nv) throw std::runtime_error(path + ": face index out of range");
                for (size_t k = 1; k + 1 < idx.size(); ++k)
                    pushTriangle(xf, &verts[3 * idx[0]], &verts[3 * idx[k]], &verts[3 * idx[k + 1]], out);
            }
        }
    }
}

}  // namespace

// MeshTriangle ctor file dispatch, src/Triangle.cpp:109-165
void loadMeshFile(const std::string& path, const Mat4& xf, std::vector<TkTriangle>* out) {
    if (path.size() < 4) throw std::runtime_error("mesh file name too short: " + path);
    std::string ext = path.substr(path.size() - 4, 4);
    if (ext == ".obj") loadObj(path, xf, out);
    else if (ext == ".ply") loadPly(path, xf, out);
    else throw std::runtime_error("unknown mesh format: " + path);
}

void loadScene(Config* config, const std::string& filename) {
    std::string text = readFile(filename);
    XmlParser parser(text);
    std::unique_ptr<XmlNode> rootPtr;
    try {
        rootPtr = parser.parseDocument();
    } catch (const std::exception& e) {
        throw std::runtime_error(filename + ": parse error: " + e.what());
    }
    const XmlNode* root = rootPtr.get();
    const std::string base = dirName(filename);

    // root attributes and image size, src/TkLoader.cpp:387-390
    attribString(root, false, "filename", &config->outfilename);
    childInt(root, false, "width", &config->width);
    childInt(root, false, "height", &config->height);
    childInt(root, false, "windows", &config->isWindows);
    // not in the reference's format: how many GPUs to spread the render over (default 1)
    childInt(root, false, "gpus", &config->gpus);
    if (config->gpus < 1 || config->gpus > 64) throw std::runtime_error(filename + ": <gpus> must be between 1 and 64");

    // parse_camera, src/TkLoader.cpp:213-226
    const XmlNode* cam = uniqueChild(root, true, "camera");
    childReal(cam, true, "fov", &config->camera.fov_degrees);
    childReal(cam, true, "near_clip", &config->camera.near_clip);
    childReal(cam, true, "far_clip", &config->camera.far_clip);
    float cp[3];
    childVec3(cam, true, "position", cp);
    config->camera.theta = cp[0];
    config->camera.phi = cp[1];
    config->camera.r = cp[2];

    // parse_tracer, src/TkLoader.cpp:228-263
    const XmlNode* tr = uniqueChild(root, true, "tracer");
    std::string type;
    attribString(tr, true, "type", &type);
    TkTracerConfig& tc = config->tracer;
    memset(&tc, 0, sizeof(tc));
    snprintf(tc.type, sizeof(tc.type), "%s", type.c_str());
    tc.spp = 16; tc.max_depth = 32; tc.num_threads = 4; tc.russian_roulette = 0.8f;
    childInt(tr, false, "spp", &tc.spp);
    childInt(tr, false, "depth", &tc.max_depth);
    childReal(tr, false, "roulete", &tc.russian_roulette);
    childInt(tr, false, "thread", &tc.num_threads);
    if (type == "sppm") {
        childInt(tr, true, "recurse", &tc.sppm.numRecurse);
        childInt(tr, true, "gsize", &tc.sppm.globalSize);
        childInt(tr, true, "gsample", &tc.sppm.globalSample);
        childInt(tr, true, "csize", &tc.sppm.causticSize);
        childInt(tr, true, "csample", &tc.sppm.causticSample);
        childReal(tr, true, "radius", &tc.sppm.radius2);
        // The reference chooses between its two SPPM classes by editing src/TkLoader.cpp:249-250; here
        // <hashgrid i="1"/> inside the tracer block selects tk::SPPM (type "sppm_hash" from here on).
        int hashgrid = 0;
        childInt(tr, false, "hashgrid", &hashgrid);
        if (hashgrid) snprintf(tc.type, sizeof(tc.type), "sppm_hash");
    }

    SceneData& sc = config->scene;
    std::map<std::string, int> materials;
    for (const XmlNode* e : root->all("material")) {
        std::string name;
        TkMaterial m = parseMaterial(e, &name);
        if (!materials.emplace(name, int(sc.materials.size())).second) failAt(e, "Material '" + name + "' multiply defined.");
        sc.materials.push_back(m);
    }

    // parse_mesh, src/TkLoader.cpp:320-328 (transform: :203-211)
    std::map<std::string, MeshEntry> meshes;
    for (const XmlNode* e : root->all("mesh")) {
        std::string name, file;
        attribString(e, true, "name", &name);
        attribString(e, false, "filename", &file);
        float t[3] = {0, 0, 0}, s[3] = {0, 0, 0}, o[3] = {0, 0, 0};
        childVec3(e, true, "position", t);
        childVec3(e, false, "orientation", o);
        childVec3(e, false, "scale", s);
        Mat4 xf = makeTransform(t, s, o);
        std::string path = base + "/" + file;
        if (!fileExists(path)) path = file;
        MeshEntry me;
        loadMeshFile(path, xf, &me.tris);
        if (!meshes.emplace(name, std::move(me)).second) failAt(e, "Mesh '" + name + "' multiply defined.");
    }

    // parse_light, src/TkLoader.cpp:330-337: one-sided area light on a mesh
    struct LightEntry { float Le[3]; std::string mesh; int index; };
    std::map<std::string, LightEntry> lights;
    for (const XmlNode* e : root->all("light")) {
        std::string name;
        LightEntry le{};
        attribString(e, true, "name", &name);
        attribString(e, false, "mesh", &le.mesh);
        if (!meshes.count(le.mesh)) failAt(e, "No such mesh '" + le.mesh + "'.");
        childColor(e, true, "color", le.Le);
        le.index = int(sc.lights.size());
        TkLight l{};
        memcpy(l.Le, le.Le, sizeof(l.Le));
        l.two_sided = 0;
        l.object = -1;
        sc.lights.push_back(l);
        if (!lights.emplace(name, le).second) failAt(e, "AreaLight '" + name + "' multiply defined.");
    }

    // parse_object, src/TkLoader.cpp:339-362
    for (const XmlNode* e : root->all("object")) {
        std::string otype, mat, mesh, light;
        attribString(e, true, "type", &otype);
        attribString(e, false, "material", &mat);
        if (!materials.count(mat)) failAt(e, "No such material '" + mat + "'.");
        TkObject ob{};
        ob.material = materials[mat];
        ob.light = -1;
        if (otype == "light" || otype == "mesh") {
            if (otype == "light") {
                attribString(e, false, "light", &light);
                if (!lights.count(light)) failAt(e, "No such light '" + light + "'.");
                ob.light = lights[light].index;
            }
            attribString(e, false, "mesh", &mesh);
            if (!meshes.count(mesh)) failAt(e, "No such mesh '" + mesh + "'.");
            const MeshEntry& me = meshes[mesh];
            ob.kind = 0;
            ob.first = int(sc.tris.size());
            ob.count = int(me.tris.size());
            sc.tris.insert(sc.tris.end(), me.tris.begin(), me.tris.end());
            if (ob.light >= 0) {
                if (lights[light].mesh != mesh) failAt(e, "light object must use the light's own mesh '" + lights[light].mesh + "'.");
                sc.lights[ob.light].object = int(sc.objects.size());
            }
        } else if (otype == "sphere") {
            TkSphere sp{};
            childVec3(e, true, "position", sp.center);
            childReal(e, true, "radius", &sp.radius);
            ob.kind = 1;
            ob.first = int(sc.spheres.size());
            ob.count = 1;
            sc.spheres.push_back(sp);
        } else {
            failAt(e, "unknown object type '" + otype + "'.");
        }
        sc.objects.push_back(ob);
    }
    for (size_t i = 0; i < sc.lights.size(); ++i)
        if (sc.lights[i].object < 0)
            throw std::runtime_error(filename + ": every <light> needs an <object type=\"light\"> that places its mesh");
}

namespace {
// pcg32 (XSH-RR 64/32), one draw per index from a stream keyed by the seed.
float pcg01(uint32_t seed, uint64_t i) {
    uint64_t state = (i + 1) * 6364136223846793005ULL + (uint64_t(seed) << 1 | 1u);
    state = state * 6364136223846793005ULL + 1442695040888963407ULL;
    uint32_t xorshifted = uint32_t(((state >> 18u) ^ state) >> 27u);
    uint32_t rot = uint32_t(state >> 59u);
    uint32_t r = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    return float(r >> 8) * (1.0f / 16777216.0f);
}
}  // namespace

void makeSyntheticScene(Config* config, const std::string& modelDir, int quads, uint32_t seed) {
    SceneData& sc = config->scene;
    sc = SceneData();
    auto addMat = [&](int type, float r, float g, float b, float eta) {
        TkMaterial m{};
        m.type = type;
        m.R[0] = r; m.R[1] = g; m.R[2] = b;
        m.T[0] = m.T[1] = m.T[2] = 1.f;
        m.etaA = 1.f; m.etaB = eta;
        sc.materials.push_back(m);
        return int(sc.materials.size()) - 1;
    };
    int red = addMat(TK_DIFFUSE, 0.63f, 0.065f, 0.05f, 1.f);
    int green = addMat(TK_DIFFUSE, 0.14f, 0.45f, 0.091f, 1.f);
    int white = addMat(TK_DIFFUSE, 0.725f, 0.71f, 0.68f, 1.f);
    int lightMat = addMat(TK_DIFFUSE, 0.65f, 0.65f, 0.65f, 1.f);
    int glass = addMat(TK_GLASS, 1.f, 1.f, 1.f, 1.75496f);
    int mirror = addMat(TK_MIRROR, 1.f, 1.f, 1.f, 1.f);
    const float t[3] = {-2.2f, -2.2f, 0.f}, s[3] = {0.008f, 0.008f, 0.008f}, o[3] = {0, 0, 0};
    Mat4 xf = makeTransform(t, s, o);
    auto addMesh = [&](const char* file, int mat, int light, size_t skipTris) {
        std::vector<TkTriangle> tris;
        loadMeshFile(modelDir + "/cornellbox/" + file, xf, &tris);
        TkObject ob{};
        ob.kind = 0;
        ob.first = int(sc.tris.size());
        ob.count = int(tris.size() - skipTris);
        ob.material = mat;
        ob.light = light;
        sc.tris.insert(sc.tris.end(), tris.begin() + long(skipTris), tris.end());
        sc.objects.push_back(ob);
        return int(sc.objects.size()) - 1;
    };
    // floor.obj holds floor, ceiling and back wall; its first two triangles are the floor
    addMesh("floor.obj", white, -1, 2);
    addMesh("left.obj", red, -1, 0);
    addMesh("right.obj", green, -1, 0);
    TkLight L{};
    L.Le[0] = 47.8348f; L.Le[1] = 38.5664f; L.Le[2] = 31.0808f;
    L.two_sided = 0;
    sc.lights.push_back(L);
    sc.lights[0].object = addMesh("light.obj", lightMat, 0, 0);
    TkSphere g{{-0.68f, -1.64f, 1.6f}, 0.56f}, mir{{0.6f, -1.4f, 2.8f}, 0.8f};
    TkObject og{1, 0, 1, glass, -1}, om{1, 1, 1, mirror, -1};
    sc.spheres.push_back(g);
    sc.spheres.push_back(mir);
    sc.objects.push_back(og);
    sc.objects.push_back(om);
    // height field over the floor rectangle
    const float x0 = -2.2f, x1 = -2.2f + 552.8f * 0.008f, z0 = 0.f, z1 = 559.2f * 0.008f;
    const int n = quads;
    std::vector<float> hy(size_t(n + 1) * size_t(n + 1));
    for (int j = 0; j <= n; ++j)
        for (int i = 0; i <= n; ++i) {
            float x = x0 + (x1 - x0) * float(i) / float(n), z = z0 + (z1 - z0) * float(j) / float(n);
            size_t id = size_t(j) * size_t(n + 1) + size_t(i);
            hy[id] = -2.2f + 0.15f * std::sin(7.f * x) * std::cos(5.f * z) + 0.02f * (pcg01(seed, id) - 0.5f);
        }
    TkObject hf{};
    hf.kind = 0;
    hf.first = int(sc.tris.size());
    hf.material = white;
    hf.light = -1;
    sc.tris.reserve(sc.tris.size() + size_t(n) * size_t(n) * 2);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            auto P = [&](int ii, int jj, float* p) {
                p[0] = x0 + (x1 - x0) * float(ii) / float(n);
                p[1] = hy[size_t(jj) * size_t(n + 1) + size_t(ii)];
                p[2] = z0 + (z1 - z0) * float(jj) / float(n);
            };
            TkTriangle a, b;
            // wound so the geometric normal points up (+y), like the Cornell floor
            P(i, j, a.v0); P(i, j + 1, a.v1); P(i + 1, j + 1, a.v2);
            P(i, j, b.v0); P(i + 1, j + 1, b.v1); P(i + 1, j, b.v2);
            sc.tris.push_back(a);
            sc.tris.push_back(b);
        }
    hf.count = int(sc.tris.size()) - hf.first;
    sc.objects.push_back(hf);

    config->camera.fov_degrees = 40.f;
    config->camera.near_clip = 0.1f;
    config->camera.far_clip = 100.f;
    config->camera.theta = -3.141592653f;
    config->camera.phi = 1.5707963265f;
    config->camera.r = 6.5f;
    memset(&config->tracer, 0, sizeof(config->tracer));
    snprintf(config->tracer.type, sizeof(config->tracer.type), "sppm");
    config->tracer.spp = 1;
    config->tracer.max_depth = 20;
    config->tracer.num_threads = 4;
    config->tracer.russian_roulette = 0.8f;
    config->tracer.sppm = TkSppmParam{64, 16000000, 100, 4000000, 50, 0.08f};
    config->width = config->height = 4096;
    config->outfilename = "synthetic.ppm";
}

namespace {
void matMul(const float a[16], const float b[16], float r[16]) {  // src/Matrix4.h:75-99
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[i * 4 + j] = a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j] + a[i * 4 + 2] * b[2 * 4 + j] +
                           a[i * 4 + 3] * b[3 * 4 + j];
}
// Matrix4::inverse, src/Matrix4.cpp:8-68
void matInverse(const float m[16], float out[16]) {
    float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    float m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    float v0 = m20 * m31 - m21 * m30, v1 = m20 * m32 - m22 * m30, v2 = m20 * m33 - m23 * m30;
    float v3 = m21 * m32 - m22 * m31, v4 = m21 * m33 - m23 * m31, v5 = m22 * m33 - m23 * m32;
    float t00 = v5 * m11 - v4 * m12 + v3 * m13;
    float t10 = -(v5 * m10 - v2 * m12 + v1 * m13);
    float t20 = v4 * m10 - v2 * m11 + v0 * m13;
    float t30 = -(v3 * m10 - v1 * m11 + v0 * m12);
    float invDet = 1.0f / (t00 * m00 + t10 * m01 + t20 * m02 + t30 * m03);
    float d00 = t00 * invDet, d10 = t10 * invDet, d20 = t20 * invDet, d30 = t30 * invDet;
    float d01 = -(v5 * m01 - v4 * m02 + v3 * m03) * invDet;
    float d11 = (v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d21 = -(v4 * m00 - v2 * m01 + v0 * m03) * invDet;
    float d31 = (v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    v0 = m10 * m31 - m11 * m30; v1 = m10 * m32 - m12 * m30; v2 = m10 * m33 - m13 * m30;
    v3 = m11 * m32 - m12 * m31; v4 = m11 * m33 - m13 * m31; v5 = m12 * m33 - m13 * m32;
    float d02 = (v5 * m01 - v4 * m02 + v3 * m03) * invDet;
    float d12 = -(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d22 = (v4 * m00 - v2 * m01 + v0 * m03) * invDet;
    float d32 = -(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    v0 = m21 * m10 - m20 * m11; v1 = m22 * m10 - m20 * m12; v2 = m23 * m10 - m20 * m13;
    v3 = m22 * m11 - m21 * m12; v4 = m23 * m11 - m21 * m13; v5 = m23 * m12 - m22 * m13;
    float d03 = -(v5 * m01 - v4 * m02 + v3 * m03) * invDet;
    float d13 = +(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d23 = -(v4 * m00 - v2 * m01 + v0 * m03) * invDet;
    float d33 = +(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    float r[16] = {d00, d01, d02, d03, d10, d11, d12, d13, d20, d21, d22, d23, d30, d31, d32, d33};
    memcpy(out, r, sizeof(r));
}
void norm3(float v[3]) {  // src/Vector.hpp:80-89
    float mag2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (mag2 > 0) {
        float inv = 1 / sqrtf(mag2);
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    }
}
void cross3(const float a[3], const float b[3], float r[3]) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
}  // namespace

void cameraMatrices(const TkCameraConfig& cam, int width, int height, float c2w[16], float raster2cam[16]) {
    // CameraInfo::rotate(0, 0), src/TkRenderer.cpp:333-356 (target = origin)
    float phi = std::fmax(std::fmin(cam.phi + 0.f, kPi), 0.f);
    float theta = cam.theta + 0.f;
    float sinPhi = float(std::sin(double(phi)));
    if (sinPhi == 0) {
        phi += 0x1p-24f;
        sinPhi = float(std::sin(double(phi)));
    }
    float r = cam.r;
    float dir[3] = {r * sinPhi * float(std::sin(double(theta))), r * float(std::cos(double(phi))),
                    r * sinPhi * float(std::cos(double(theta)))};
    float pos[3] = {0.f + dir[0], 0.f + dir[1], 0.f + dir[2]};
    float up[3] = {0, sinPhi > 0 ? 1.f : -1.f, 0};
    float right[3], newUp[3], toward[3] = {dir[0], dir[1], dir[2]};
    cross3(up, dir, right);
    norm3(right);
    cross3(dir, right, newUp);
    norm3(newUp);
    norm3(toward);
    float m[16] = {right[0], newUp[0], toward[0], pos[0],   // setTranslation: src/main.cpp:40
                   right[1], newUp[1], toward[1], pos[1],
                   right[2], newUp[2], toward[2], pos[2],
                   0, 0, 0, 1};
    memcpy(c2w, m, sizeof(m));

    // CameraInfo::perspective, src/TkRenderer.cpp:358-376
    float c2s[16] = {0};
    float fovRad = cam.fov_degrees * (kPi / 180.0f);
    float h = 1.0f / float(std::tan(double(fovRad * 0.5f)));
    float w = h * float(height) / float(width);
    float fa = cam.far_clip, ne = cam.near_clip;
    c2s[0] = w;
    c2s[5] = h;
    c2s[10] = -fa / (fa - ne);
    c2s[11] = -fa * ne / (fa - ne);
    c2s[14] = -1.0f;
    // Camera ctor, src/core/camera.cpp:14-20
    float s2r[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    s2r[0] = width * 0.5f; s2r[5] = height * 0.5f; s2r[10] = 1;
    s2r[3] = width * 0.5f; s2r[7] = height * 0.5f; s2r[11] = 0;
    float c2r[16];
    matMul(s2r, c2s, c2r);
    matInverse(c2r, raster2cam);
}

}  // namespace tk
