// See TorchArchive.h.  Three small pieces: a ZIP container (stored records only), the subset of pickle protocol 2 that LibTorch's pickler
// emits, and the two layouts the reference writes (an Agent module tree, an AdamW optimizer).
#include "Utils/TorchArchive.h"

#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>

namespace ppo {
namespace pt {
namespace {

[[noreturn]] void fail(const std::string& file, const std::string& what) { throw std::runtime_error("checkpoint " + file + ": " + what); }

// ------------------------------------------------------------------ CRC-32 (ZIP polynomial) -------------------------------------------------
uint32_t crc32(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ------------------------------------------------------------------ ZIP, reading --------------------------------------------------------------
struct ZipEntry { uint16_t method; uint32_t crc, csize, usize, local; };
class ZipReader {
public:
    explicit ZipReader(const std::string& path) : file_(path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) fail(path, "cannot be opened");
        f.seekg(0, std::ios::end);
        const std::streamoff len = f.tellg();
        if (len < 22) fail(path, "is too short to be a ZIP archive");
        if (len > (std::streamoff)1 << 31) fail(path, "is larger than 2 GiB (ZIP64 is not read)");
        bytes_.resize((size_t)len);
        f.seekg(0);
        f.read(reinterpret_cast<char*>(bytes_.data()), len);
        if (!f) fail(path, "could not be read");
        // end-of-central-directory record: the last 22 bytes, or earlier when a comment follows it
        size_t eocd = std::string::npos;
        const size_t lowest = bytes_.size() > 22 + 65535 ? bytes_.size() - 22 - 65535 : 0;
        for (size_t i = bytes_.size() - 22 + 1; i-- > lowest;) {
            if (rd32(&bytes_[i]) == 0x06054b50u) { eocd = i; break; }
        }
        if (eocd == std::string::npos) fail(path, "has no ZIP end-of-central-directory record (truncated?)");
        const uint32_t count = rd16(&bytes_[eocd + 10]), cd_size = rd32(&bytes_[eocd + 12]), cd_off = rd32(&bytes_[eocd + 16]);
        if (count == 0xFFFFu || cd_off == 0xFFFFFFFFu) fail(path, "is a ZIP64 archive (not read)");
        if ((uint64_t)cd_off + cd_size > eocd) fail(path, "has a central directory that runs past the end of the file");
        size_t p = cd_off;
        for (uint32_t i = 0; i < count; i++) {
            if (p + 46 > eocd || rd32(&bytes_[p]) != 0x02014b50u) fail(path, "has a damaged central directory");
            const uint16_t nl = rd16(&bytes_[p + 28]), el = rd16(&bytes_[p + 30]), cl = rd16(&bytes_[p + 32]);
            if (p + 46 + nl > eocd) fail(path, "has a damaged central directory");
            ZipEntry e;
            e.method = rd16(&bytes_[p + 10]); e.crc = rd32(&bytes_[p + 16]); e.csize = rd32(&bytes_[p + 20]); e.usize = rd32(&bytes_[p + 24]);
            e.local = rd32(&bytes_[p + 42]);
            const std::string name(reinterpret_cast<const char*>(&bytes_[p + 46]), nl);
            entries_[name] = e;
            order_.push_back(name);
            p += 46u + nl + el + cl;
        }
    }
    const std::vector<std::string>& names() const { return order_; }
    bool has(const std::string& name) const { return entries_.count(name) != 0; }
    // the bytes of a stored record, CRC checked
    std::pair<const uint8_t*, size_t> record(const std::string& name) const {
        const auto it = entries_.find(name);
        if (it == entries_.end()) fail(file_, "has no record " + name);
        const ZipEntry& e = it->second;
        if (e.method != 0) fail(file_, "record " + name + " is compressed (method " + std::to_string(e.method) + "); only stored records are read");
        if ((size_t)e.local + 30 > bytes_.size() || rd32(&bytes_[e.local]) != 0x04034b50u) fail(file_, "record " + name + " has a damaged local header");
        const size_t data = (size_t)e.local + 30 + rd16(&bytes_[e.local + 26]) + rd16(&bytes_[e.local + 28]);
        if (data + e.usize > bytes_.size()) fail(file_, "record " + name + " runs past the end of the file (truncated?)");
        if (crc32(&bytes_[data], e.usize) != e.crc) fail(file_, "record " + name + " fails its CRC check");
        return { &bytes_[data], (size_t)e.usize };
    }
private:
    std::string file_;
    std::vector<uint8_t> bytes_;
    std::map<std::string, ZipEntry> entries_;
    std::vector<std::string> order_;
};

// ------------------------------------------------------------------ ZIP, writing ----------------------------------------------------------------
class ZipWriter {
public:
    void add(const std::string& name, const void* data, size_t n) {
        Rec r;
        r.name = name; r.offset = (uint32_t)out_.size(); r.size = (uint32_t)n; r.crc = crc32(static_cast<const uint8_t*>(data), n);
        // tensor storages start on a 64-byte boundary, as LibTorch's writer places them (its reader maps them in place): the slack goes into
        // an extra field of the local header
        const size_t header = 30 + name.size();
        size_t pad = (64 - (out_.size() + header + 4) % 64) % 64;
        const uint16_t extra = (uint16_t)(4 + pad);
        put32(0x04034b50u); put16(20); put16(0x0800); put16(0); put16(0); put16(0x21);   // version, flags (UTF-8 names), stored, time, date
        put32(r.crc); put32(r.size); put32(r.size); put16((uint16_t)name.size()); put16(extra);
        out_.insert(out_.end(), name.begin(), name.end());
        put16(0x4246); put16((uint16_t)pad);                                               // "FB" padding field
        out_.insert(out_.end(), pad, (uint8_t)'Z');
        const uint8_t* b = static_cast<const uint8_t*>(data);
        out_.insert(out_.end(), b, b + n);
        recs_.push_back(r);
    }
    void add(const std::string& name, const std::string& text) { add(name, text.data(), text.size()); }
    void finish(const std::string& path) {
        const uint32_t cd = (uint32_t)out_.size();
        for (const Rec& r : recs_) {
            put32(0x02014b50u); put16(20); put16(20); put16(0x0800); put16(0); put16(0); put16(0x21);
            put32(r.crc); put32(r.size); put32(r.size); put16((uint16_t)r.name.size()); put16(0); put16(0); put16(0); put16(0); put32(0);
            put32(r.offset);
            out_.insert(out_.end(), r.name.begin(), r.name.end());
        }
        const uint32_t cd_size = (uint32_t)out_.size() - cd;
        put32(0x06054b50u); put16(0); put16(0); put16((uint16_t)recs_.size()); put16((uint16_t)recs_.size()); put32(cd_size); put32(cd); put16(0);
        std::ofstream f(path, std::ios::binary);
        f.write(reinterpret_cast<const char*>(out_.data()), (std::streamsize)out_.size());
        f.flush();
        if (!f.good()) throw std::runtime_error("could not write checkpoint " + path);
    }
private:
    struct Rec { std::string name; uint32_t offset, size, crc; };
    void put16(uint16_t v) { out_.push_back((uint8_t)v); out_.push_back((uint8_t)(v >> 8)); }
    void put32(uint32_t v) { for (int i = 0; i < 4; i++) out_.push_back((uint8_t)(v >> (8 * i))); }
    std::vector<uint8_t> out_;
    std::vector<Rec> recs_;
};

// ------------------------------------------------------------------ pickle, reading -----------------------------------------------------------
struct Value;
using V = std::shared_ptr<Value>;
struct Value {
    enum Kind { NONE, BOOL, INT, FLOAT, STR, TUPLE, LIST, DICT, GLOBAL, OBJECT, PERSID, REDUCED, MARK } kind = NONE;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;                          // STR: the text; GLOBAL / OBJECT: "module name" of the class
    std::vector<V> items;                   // TUPLE, LIST; PERSID: the id tuple's items; REDUCED: { callable, argument tuple }
    std::vector<std::pair<V, V>> dict;      // DICT; OBJECT: its state, in insertion order
};
V mk(Value::Kind k) { auto v = std::make_shared<Value>(); v->kind = k; return v; }

class Unpickler {
public:
    Unpickler(const std::string& file, const uint8_t* p, size_t n) : file_(file), p_(p), n_(n) {}
    V run() {
        for (;;) {
            const uint8_t op = u8();
            switch (op) {
            case 0x80: u8(); break;                                                      // PROTO
            case '.': if (stack_.empty()) bad("STOP on an empty stack"); return stack_.back();
            case 'N': stack_.push_back(mk(Value::NONE)); break;
            case 0x88: case 0x89: { V v = mk(Value::BOOL); v->b = op == 0x88; stack_.push_back(v); break; }
            case 'K': pushInt(u8()); break;
            case 'M': { const uint8_t* q = take(2); pushInt(rd16(q)); break; }
            case 'J': { const uint8_t* q = take(4); pushInt((int32_t)rd32(q)); break; }
            case 0x8a: {                                                                 // LONG1: little-endian two's complement, n bytes
                const uint8_t n = u8();
                if (n > 8) bad("integer wider than 64 bits");
                const uint8_t* q = take(n);
                uint64_t u = 0;
                for (int k = 0; k < n; k++) u |= (uint64_t)q[k] << (8 * k);
                if (n > 0 && n < 8 && (q[n - 1] & 0x80)) u |= ~(uint64_t)0 << (8 * n);
                pushInt((int64_t)u);
                break;
            }
            case 'G': {                                                                  // BINFLOAT: big-endian binary64
                const uint8_t* q = take(8);
                uint64_t u = 0;
                for (int k = 0; k < 8; k++) u = (u << 8) | q[k];
                V v = mk(Value::FLOAT);
                std::memcpy(&v->f, &u, 8);
                stack_.push_back(v);
                break;
            }
            case 'X': { const uint32_t n = rd32(take(4)); V v = mk(Value::STR); v->s.assign(reinterpret_cast<const char*>(take(n)), n); stack_.push_back(v); break; }
            case 'c': { V v = mk(Value::GLOBAL); v->s = line() + " "; v->s += line(); stack_.push_back(v); break; }
            case 'q': memo_[u8()] = top(); break;
            case 'r': memo_[rd32(take(4))] = top(); break;
            case 'h': stack_.push_back(memo(u8())); break;
            case 'j': stack_.push_back(memo(rd32(take(4)))); break;
            case '(': stack_.push_back(mk(Value::MARK)); break;
            case ')': stack_.push_back(mk(Value::TUPLE)); break;
            case '}': stack_.push_back(mk(Value::DICT)); break;
            case ']': stack_.push_back(mk(Value::LIST)); break;
            case 't': { V v = mk(Value::TUPLE); v->items = popToMark(); stack_.push_back(v); break; }
            case 0x85: case 0x86: case 0x87: {
                const size_t n = op - 0x84;
                if (stack_.size() < n) bad("tuple on a short stack");
                V v = mk(Value::TUPLE);
                v->items.assign(stack_.end() - (std::ptrdiff_t)n, stack_.end());
                stack_.resize(stack_.size() - n);
                stack_.push_back(v);
                break;
            }
            case 'l': { V v = mk(Value::LIST); v->items = popToMark(); stack_.push_back(v); break; }
            case 'a': { V x = pop(); V l = top(); if (l->kind != Value::LIST) bad("APPEND to a non-list"); l->items.push_back(x); break; }
            case 'e': { std::vector<V> xs = popToMark(); V l = top(); if (l->kind != Value::LIST) bad("APPENDS to a non-list"); l->items.insert(l->items.end(), xs.begin(), xs.end()); break; }
            case 's': { V val = pop(); V key = pop(); V d = top(); if (d->kind != Value::DICT) bad("SETITEM on a non-dict"); d->dict.emplace_back(key, val); break; }
            case 'u': {
                std::vector<V> xs = popToMark();
                V d = top();
                if (d->kind != Value::DICT || xs.size() % 2) bad("SETITEMS on a non-dict");
                for (size_t k = 0; k < xs.size(); k += 2) d->dict.emplace_back(xs[k], xs[k + 1]);
                break;
            }
            case 'Q': { V id = pop(); V v = mk(Value::PERSID); if (id->kind == Value::TUPLE) v->items = id->items; else v->items.push_back(id); stack_.push_back(v); break; }
            case 0x81: { V args = pop(); V cls = pop(); if (cls->kind != Value::GLOBAL) bad("NEWOBJ of a non-class"); V v = mk(Value::OBJECT); v->s = cls->s; (void)args; stack_.push_back(v); break; }
            case 'R': { V args = pop(); V fn = pop(); V v = mk(Value::REDUCED); v->items = { fn, args }; stack_.push_back(v); break; }
            case 'b': {
                V state = pop();
                V obj = top();
                if (obj->kind != Value::OBJECT) bad("BUILD on a non-object");
                if (state->kind == Value::DICT) obj->dict = state->dict;
                else obj->items.push_back(state);
                break;
            }
            default: bad("pickle opcode 0x" + hex(op) + " is not one LibTorch's module pickler emits");
            }
        }
    }
private:
    static std::string hex(uint8_t b) { const char* d = "0123456789abcdef"; return std::string(1, d[b >> 4]) + d[b & 15]; }
    [[noreturn]] void bad(const std::string& what) const { fail(file_, "data.pkl at byte " + std::to_string(at_) + ": " + what); }
    const uint8_t* take(size_t n) { if (at_ + n > n_) bad("runs past its end"); const uint8_t* q = p_ + at_; at_ += n; return q; }
    uint8_t u8() { return *take(1); }
    std::string line() { std::string s; for (;;) { const char c = (char)u8(); if (c == '\n') return s; s += c; } }
    void pushInt(int64_t i) { V v = mk(Value::INT); v->i = i; stack_.push_back(v); }
    V top() { if (stack_.empty() || stack_.back()->kind == Value::MARK) bad("empty stack"); return stack_.back(); }
    V pop() { V v = top(); stack_.pop_back(); return v; }
    V memo(uint32_t k) { const auto it = memo_.find(k); if (it == memo_.end()) bad("reads memo slot " + std::to_string(k) + " before writing it"); return it->second; }
    std::vector<V> popToMark() {
        size_t k = stack_.size();
        while (k > 0 && stack_[k - 1]->kind != Value::MARK) k--;
        if (k == 0) bad("no MARK on the stack");
        std::vector<V> xs(stack_.begin() + (std::ptrdiff_t)k, stack_.end());
        stack_.resize(k - 1);
        return xs;
    }
    std::string file_;
    const uint8_t* p_;
    size_t n_, at_ = 0;
    std::vector<V> stack_;
    std::map<uint32_t, V> memo_;
};

// ------------------------------------------------------------------ the module tree ---------------------------------------------------------
struct Archive {
    std::string file, stem;
    ZipReader zip;
    V root;
    explicit Archive(const std::string& path) : file(path), zip(path) {
        for (const std::string& n : zip.names()) {
            const size_t k = n.rfind("/data.pkl");
            if (k != std::string::npos && k + 9 == n.size() && n.find('/') == k) { stem = n.substr(0, k); break; }
        }
        if (stem.empty()) fail(path, "holds no <name>/data.pkl record: not a LibTorch module archive");
        const auto rec = zip.record(stem + "/data.pkl");
        root = Unpickler(path, rec.first, rec.second).run();
        if (root->kind != Value::OBJECT) fail(path, "data.pkl does not describe a module object");
    }
    static V find(const V& obj, const std::string& key) {
        for (const auto& kv : obj->dict) if (kv.first->kind == Value::STR && kv.first->s == key) return kv.second;
        return nullptr;
    }
    bool isTensor(const V& v) const {
        return v && v->kind == Value::REDUCED && v->items[0]->kind == Value::GLOBAL && v->items[0]->s.rfind("torch._utils _rebuild_tensor", 0) == 0;
    }
    // _rebuild_tensor_v2(storage = persistent id ('storage', torch.<T>Storage, key, device, numel), offset, sizes, strides, requires_grad, hooks)
    NamedTensor tensor(const V& v, const std::string& name) const {
        const V& args = v->items[1];
        if (args->kind != Value::TUPLE || args->items.size() < 4) fail(file, name + ": unexpected tensor record");
        const V& st = args->items[0];
        if (st->kind != Value::PERSID || st->items.size() < 5 || st->items[1]->kind != Value::GLOBAL || st->items[2]->kind != Value::STR)
            fail(file, name + ": unexpected storage record");
        if (st->items[1]->s != "torch FloatStorage") fail(file, name + ": storage type " + st->items[1]->s + " (float32 expected)");
        const auto rec = zip.record(stem + "/data/" + st->items[2]->s);
        // every field below comes out of the file: check its kind before its value is read, and bound the element count by what the storage record
        // can hold BEFORE anything is sized by it (a damaged or hostile archive must end in "ignored, because ...", not in bad_alloc or an overflow)
        if (args->items[1]->kind != Value::INT || args->items[2]->kind != Value::TUPLE || args->items[3]->kind != Value::TUPLE)
            fail(file, name + ": unexpected tensor record (offset / sizes / strides)");
        const int64_t storage_numel = (int64_t)(rec.second / 4), offset = args->items[1]->i;
        NamedTensor t;
        t.name = name;
        std::vector<int64_t> strides;
        for (const V& d : args->items[2]->items) { if (d->kind != Value::INT) fail(file, name + ": a size is not an integer"); t.sizes.push_back(d->i); }
        for (const V& d : args->items[3]->items) { if (d->kind != Value::INT) fail(file, name + ": a stride is not an integer"); strides.push_back(d->i); }
        if (strides.size() != t.sizes.size()) fail(file, name + ": sizes and strides disagree");
        if (t.sizes.size() > 8) fail(file, name + ": more than 8 dimensions");
        int64_t numel = 1;
        for (int64_t d : t.sizes) {
            if (d < 0) fail(file, name + ": negative size");
            if (d != 0 && numel > storage_numel / d) fail(file, name + ": more elements than its storage record holds");   // also rules out overflow
            numel *= d;
        }
        if (offset < 0 || numel > storage_numel) fail(file, name + ": more elements than its storage record holds");
        // a stride no honest view needs (every index x stride product then stays far inside int64: the element loop cannot overflow before its bound check)
        // PyTorch never dereferences the stride of a size-1 (or size-0) dimension and records 1-strides for empty tensors: only dimensions that step are bound
        for (size_t d = 0; numel > 0 && d < strides.size(); d++)
            if (t.sizes[d] > 1 && (strides[d] < -storage_numel || strides[d] > storage_numel)) fail(file, name + ": stride outside its storage");
        if (offset > storage_numel) fail(file, name + ": offset outside its storage");
        t.values.resize((size_t)numel);
        std::vector<int64_t> idx(t.sizes.size(), 0);
        for (int64_t e = 0; e < numel; e++) {
            int64_t at = offset;
            for (size_t d = 0; d < idx.size(); d++) at += idx[d] * strides[d];
            if (at < 0 || at >= storage_numel) fail(file, name + ": element outside its storage");
            std::memcpy(&t.values[(size_t)e], rec.first + 4 * at, 4);
            for (size_t d = idx.size(); d-- > 0;) { if (++idx[d] < t.sizes[d]) break; idx[d] = 0; }
        }
        return t;
    }
    void collect(const V& obj, const std::string& prefix, std::vector<NamedTensor>& out, int depth = 0) const {
        if (depth > 32) fail(file, "module tree deeper than 32 levels (a memo self-reference?)");
        for (const auto& kv : obj->dict) {
            if (kv.first->kind != Value::STR) continue;
            const std::string name = prefix.empty() ? kv.first->s : prefix + "." + kv.first->s;
            if (isTensor(kv.second)) out.push_back(tensor(kv.second, name));
            else if (kv.second->kind == Value::OBJECT) collect(kv.second, name, out, depth + 1);
        }
    }
};

// ------------------------------------------------------------------ pickle, writing -------------------------------------------------------------
class Pickler {
public:
    Pickler() { b_.push_back(0x80); b_.push_back(2); }
    void global(const std::string& module, const std::string& name) { b_.push_back('c'); text(module + "\n" + name + "\n"); }
    void str(const std::string& s) { b_.push_back('X'); put32((uint32_t)s.size()); text(s); }
    void integer(int64_t v) {
        if (v >= 0 && v < 256) { b_.push_back('K'); b_.push_back((uint8_t)v); }
        else if (v >= 0 && v < 65536) { b_.push_back('M'); b_.push_back((uint8_t)v); b_.push_back((uint8_t)(v >> 8)); }
        else if (v >= INT32_MIN && v <= INT32_MAX) { b_.push_back('J'); put32((uint32_t)(int32_t)v); }
        else { b_.push_back(0x8a); b_.push_back(8); for (int k = 0; k < 8; k++) b_.push_back((uint8_t)((uint64_t)v >> (8 * k))); }
    }
    void real(double v) { uint64_t u; std::memcpy(&u, &v, 8); b_.push_back('G'); for (int k = 7; k >= 0; k--) b_.push_back((uint8_t)(u >> (8 * k))); }
    void boolean(bool v) { b_.push_back(v ? 0x88 : 0x89); }
    void op(uint8_t o) { b_.push_back(o); }
    // `class(*())` with an empty state dictionary left open for the caller's items: GLOBAL, EMPTY_TUPLE, NEWOBJ, EMPTY_DICT, MARK
    void beginObject(const std::string& cls) { global(cls.substr(0, cls.rfind('.')), cls.substr(cls.rfind('.') + 1)); op(')'); op(0x81); op('}'); op('('); }
    void endObject() { op('u'); op('b'); }
    void tensor(const std::string& key, const std::vector<int64_t>& sizes, bool requires_grad, bool long_storage = false) {
        int64_t numel = 1;
        for (int64_t d : sizes) numel *= d;
        global("torch._utils", "_rebuild_tensor_v2");
        op('('); op('(');
        str("storage"); global("torch", long_storage ? "LongStorage" : "FloatStorage"); str(key); str("cpu"); integer(numel);
        op('t'); op('Q');
        integer(0);
        op('('); for (int64_t d : sizes) integer(d); op('t');
        op('(');
        for (size_t d = 0; d < sizes.size(); d++) { int64_t st = 1; for (size_t k = d + 1; k < sizes.size(); k++) st *= sizes[k]; integer(st); }
        op('t');
        boolean(requires_grad);
        global("collections", "OrderedDict"); op(')'); op('R');
        op('t'); op('R');
    }
    std::string finish() { b_.push_back('.'); return std::string(b_.begin(), b_.end()); }
private:
    void text(const std::string& s) { b_.insert(b_.end(), s.begin(), s.end()); }
    void put32(uint32_t v) { for (int k = 0; k < 4; k++) b_.push_back((uint8_t)(v >> (8 * k))); }
    std::vector<uint8_t> b_;
};

std::string stemOf(const std::string& path) {
    const size_t slash = path.find_last_of("/\\");
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.rfind('.');
    if (dot != std::string::npos && dot > 0) name.resize(dot);
    return name.empty() ? "archive" : name;
}
// class k of an archive: the root is __torch__.Module, every further one __torch__.___torch_mangle_<k-1>.Module, numbered in the order the
// objects are first met (depth first)
std::string className(int k) { return k == 0 ? "__torch__.Module" : "__torch__.___torch_mangle_" + std::to_string(k - 1) + ".Module"; }
std::string codePath(const std::string& stem, int k) {
    return k == 0 ? stem + "/code/__torch__.py" : stem + "/code/__torch__/___torch_mangle_" + std::to_string(k - 1) + ".py";
}
bool plainIdentifier(const std::string& s) {
    if (s.empty() || (s[0] >= '0' && s[0] <= '9')) return false;
    for (char c : s) if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_')) return false;
    return true;
}
// TorchScript declaration of one module class: attribute names and types in the order of the state dictionary
std::string classSource(const std::vector<std::string>& parameters, const std::vector<std::pair<std::string, std::string>>& attributes) {
    std::string s = "class Module(Module):\n  __parameters__ = [";
    for (const std::string& p : parameters) s += "\"" + p + "\", ";
    s += "]\n  __buffers__ = []\n";
    bool annotated = false;
    for (const auto& a : attributes) annotated = annotated || !plainIdentifier(a.first);
    if (annotated) s += "  __annotations__ = []\n";
    for (const auto& a : attributes) {
        if (annotated && !plainIdentifier(a.first)) s += "  __annotations__[\"" + a.first + "\"] = " + a.second + "\n";
    }
    for (const auto& a : attributes) {
        if (plainIdentifier(a.first)) s += "  " + a.first + " : " + a.second + "\n";
    }
    return s;
}
void closeArchive(ZipWriter& zip, const std::string& stem, const std::string& path) {
    zip.add(stem + "/constants.pkl", std::string("\x80\x02).", 4));
    zip.add(stem + "/version", std::string("3\n"));
    zip.add(stem + "/byteorder", std::string("little"));
    zip.finish(path);
}

}  // namespace

bool isTorchArchive(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    char m[4] = {};
    f.read(m, 4);
    return f && m[0] == 'P' && m[1] == 'K' && m[2] == 3 && m[3] == 4;
}

std::vector<std::vector<int64_t>> agentShapes(int64_t obs, int64_t hidden, int64_t act) {
    return { { hidden, obs }, { hidden }, { hidden, hidden }, { hidden }, { 1, hidden }, { 1 },
             { hidden, obs }, { hidden }, { hidden, hidden }, { hidden }, { act, hidden }, { act } };
}

AgentFile readAgent(const std::string& path) {
    Archive a(path);
    AgentFile out;
    a.collect(a.root, "", out.tensors);
    if (out.tensors.empty()) fail(path, "holds no tensors");
    return out;
}

OptimizerFile readOptimizer(const std::string& path) {
    Archive a(path);
    OptimizerFile out;
    const V version = Archive::find(a.root, "pytorch_version"), state = Archive::find(a.root, "state"), groups = Archive::find(a.root, "param_groups");
    if (!version || version->kind != Value::STR || !state || state->kind != Value::OBJECT || !groups || groups->kind != Value::OBJECT)
        fail(path, "is not an optimizer archive of the pytorch_version / state / param_groups layout (LibTorch >= 1.5)");
    const V group = Archive::find(groups, "param_groups/0");
    if (!group || group->kind != Value::OBJECT) fail(path, "has no parameter group");
    if (Archive::find(groups, "param_groups/1")) fail(path, "has more than one parameter group (the reference builds one, PPO_Discrete.cpp:76-78)");
    for (int i = 0;; i++) {
        const V key = Archive::find(group, "params/" + std::to_string(i));
        if (!key) break;
        if (key->kind != Value::STR) fail(path, "parameter key " + std::to_string(i) + " is not a string");
        const V st = Archive::find(state, key->s);
        if (!st || st->kind != Value::OBJECT) fail(path, "has no state for parameter " + std::to_string(i) + " (no optimizer step was taken before it was saved?)");
        const V step = Archive::find(st, "step"), m = Archive::find(st, "exp_avg"), v = Archive::find(st, "exp_avg_sq");
        if (!step || step->kind != Value::INT || !a.isTensor(m) || !a.isTensor(v)) fail(path, "state of parameter " + std::to_string(i) + " is not AdamW's (step, exp_avg, exp_avg_sq)");
        out.step.push_back(step->i);
        out.exp_avg.push_back(a.tensor(m, "exp_avg/" + std::to_string(i)));
        out.exp_avg_sq.push_back(a.tensor(v, "exp_avg_sq/" + std::to_string(i)));
    }
    if (out.step.empty()) fail(path, "lists no parameters");
    const V opt = Archive::find(group, "options");
    if (opt && opt->kind == Value::OBJECT) {
        auto num = [&](const char* k, double& dst) { const V x = Archive::find(opt, k); if (x && x->kind == Value::FLOAT) dst = x->f; };
        num("lr", out.lr); num("eps", out.eps); num("weight_decay", out.weight_decay);
        const V betas = Archive::find(opt, "betas");
        if (betas && betas->kind == Value::TUPLE && betas->items.size() == 2) { out.beta1 = betas->items[0]->f; out.beta2 = betas->items[1]->f; }
        const V ams = Archive::find(opt, "amsgrad");
        if (ams && ams->kind == Value::BOOL) out.amsgrad = ams->b;
    }
    return out;
}

void writeAgent(const std::string& path, int64_t obs, int64_t hidden, int64_t act, const std::vector<float>& flat, const std::string& stem_in) {
    const auto shapes = agentShapes(obs, hidden, act);
    int64_t total = 0;
    for (const auto& s : shapes) { int64_t n = 1; for (int64_t d : s) n *= d; total += n; }
    if ((int64_t)flat.size() != total) throw std::runtime_error("writeAgent: " + std::to_string(flat.size()) + " values for " + std::to_string(total) + " parameters");
    const std::string stem = stem_in.empty() ? stemOf(path) : stem_in;
    ZipWriter zip;
    Pickler p;
    // storages first (LibTorch writes them before data.pkl), then the object tree; classes are numbered as they are met
    size_t at = 0;
    for (size_t i = 0; i < shapes.size(); i++) {
        int64_t n = 1;
        for (int64_t d : shapes[i]) n *= d;
        zip.add(stem + "/data/" + std::to_string(i), flat.data() + at, (size_t)n * 4);
        at += (size_t)n;
    }
    std::vector<std::pair<int, std::string>> sources;   // class number -> declaration
    int next_class = 0, next_tensor = 0;
    const char* nets[2] = { "m_Critic", "m_Actor" };
    const char* layers[2][3] = { { "criticInputLayer", "criticMiddleLayer", "criticOutputLayer" }, { "actorInputLayer", "actorMiddleLayer", "actorOutputLayer" } };
    const int root = next_class++;
    std::vector<std::pair<std::string, std::string>> root_attrs;
    p.beginObject(className(root));
    for (int net = 0; net < 2; net++) {
        const int nc = next_class++;
        root_attrs.emplace_back(nets[net], className(nc));
        std::vector<std::pair<std::string, std::string>> net_attrs;
        p.str(nets[net]);
        p.beginObject(className(nc));
        for (int l = 0; l < 3; l++) {
            const int lc = next_class++;
            net_attrs.emplace_back(layers[net][l], className(lc));
            p.str(layers[net][l]);
            p.beginObject(className(lc));
            p.str("weight"); p.tensor(std::to_string(next_tensor), shapes[(size_t)next_tensor], true); next_tensor++;
            p.str("bias"); p.tensor(std::to_string(next_tensor), shapes[(size_t)next_tensor], true); next_tensor++;
            p.endObject();
            sources.emplace_back(lc, classSource({ "weight", "bias" }, { { "weight", "Tensor" }, { "bias", "Tensor" } }));
            if (l < 2) {   // the activation modules between the layers (Agent.cpp:27-31, 46-50): no state
                const int tc = next_class++;
                const std::string tn = "Tanh" + std::to_string(l + 1);
                net_attrs.emplace_back(tn, className(tc));
                p.str(tn);
                p.beginObject(className(tc));
                p.endObject();
                sources.emplace_back(tc, classSource({}, {}));
            }
        }
        p.endObject();
        sources.emplace_back(nc, classSource({}, net_attrs));
    }
    p.endObject();
    sources.emplace_back(root, classSource({}, root_attrs));
    zip.add(stem + "/data.pkl", p.finish());
    for (int k = 0; k < next_class; k++)
        for (const auto& s : sources) if (s.first == k) zip.add(codePath(stem, k), s.second);
    closeArchive(zip, stem, path);
}

void writeOptimizer(const std::string& path, int64_t obs, int64_t hidden, int64_t act, const std::vector<float>& exp_avg,
                    const std::vector<float>& exp_avg_sq, int64_t step, double lr, double eps, double weight_decay, const std::string& stem_in) {
    const auto shapes = agentShapes(obs, hidden, act);
    int64_t total = 0;
    for (const auto& s : shapes) { int64_t n = 1; for (int64_t d : s) n *= d; total += n; }
    if ((int64_t)exp_avg.size() != total || (int64_t)exp_avg_sq.size() != total) throw std::runtime_error("writeOptimizer: moment vectors do not match the parameter count");
    const std::string stem = stem_in.empty() ? stemOf(path) : stem_in;
    ZipWriter zip;
    Pickler p;
    const int n_params = (int)shapes.size();
    // LibTorch keys a parameter's state by the address of its tensor; any distinct strings serve (its loader maps them back by position)
    std::vector<std::string> keys;
    for (int i = 0; i < n_params; i++) keys.push_back(std::to_string(1000 + i));
    size_t at = 0;
    for (int i = 0; i < n_params; i++) {
        int64_t n = 1;
        for (int64_t d : shapes[(size_t)i]) n *= d;
        zip.add(stem + "/data/" + std::to_string(2 * i), exp_avg.data() + at, (size_t)n * 4);
        zip.add(stem + "/data/" + std::to_string(2 * i + 1), exp_avg_sq.data() + at, (size_t)n * 4);
        at += (size_t)n;
    }
    const int64_t one = 1, count = n_params;
    zip.add(stem + "/data/" + std::to_string(2 * n_params), &one, 8);          // param_groups/size
    zip.add(stem + "/data/" + std::to_string(2 * n_params + 1), &count, 8);    // params/size
    std::vector<std::pair<int, std::string>> sources;
    int next_class = 0;
    const int root = next_class++;
    p.beginObject(className(root));
    p.str("pytorch_version"); p.str("1.5.0");
    const int sc = next_class++;
    p.str("state");
    p.beginObject(className(sc));
    std::vector<std::pair<std::string, std::string>> state_attrs;
    for (int i = 0; i < n_params; i++) {
        const int pc = next_class++;
        state_attrs.emplace_back(keys[(size_t)i], className(pc));
        p.str(keys[(size_t)i]);
        p.beginObject(className(pc));
        p.str("step"); p.integer(step);
        p.str("exp_avg"); p.tensor(std::to_string(2 * i), shapes[(size_t)i], false);
        p.str("exp_avg_sq"); p.tensor(std::to_string(2 * i + 1), shapes[(size_t)i], false);
        p.endObject();
        sources.emplace_back(pc, classSource({}, { { "step", "int" }, { "exp_avg", "Tensor" }, { "exp_avg_sq", "Tensor" } }));
    }
    p.endObject();
    sources.emplace_back(sc, classSource({}, state_attrs));
    const int gc = next_class++;
    p.str("param_groups");
    p.beginObject(className(gc));
    p.str("param_groups/size"); p.tensor(std::to_string(2 * n_params), {}, false, true);
    const int g0 = next_class++;
    p.str("param_groups/0");
    p.beginObject(className(g0));
    std::vector<std::pair<std::string, std::string>> g0_attrs = { { "params/size", "Tensor" } };
    p.str("params/size"); p.tensor(std::to_string(2 * n_params + 1), {}, false, true);
    for (int i = 0; i < n_params; i++) {
        g0_attrs.emplace_back("params/" + std::to_string(i), "str");
        p.str("params/" + std::to_string(i)); p.str(keys[(size_t)i]);
    }
    const int oc = next_class++;
    g0_attrs.emplace_back("options", className(oc));
    p.str("options");
    p.beginObject(className(oc));
    p.str("lr"); p.real(lr);
    p.str("betas"); p.real(0.9); p.real(0.999); p.op(0x86);
    p.str("eps"); p.real(eps);
    p.str("weight_decay"); p.real(weight_decay);
    p.str("amsgrad"); p.boolean(false);
    p.endObject();
    sources.emplace_back(oc, classSource({}, { { "lr", "float" }, { "betas", "Tuple[float, float]" }, { "eps", "float" }, { "weight_decay", "float" }, { "amsgrad", "bool" } }));
    p.endObject();
    sources.emplace_back(g0, classSource({ "params/size" }, g0_attrs));
    p.endObject();
    sources.emplace_back(gc, classSource({ "param_groups/size" }, { { "param_groups/size", "Tensor" }, { "param_groups/0", className(g0) } }));
    p.endObject();
    sources.emplace_back(root, classSource({}, { { "pytorch_version", "str" }, { "state", className(sc) }, { "param_groups", className(gc) } }));
    zip.add(stem + "/data.pkl", p.finish());
    for (int k = 0; k < next_class; k++)
        for (const auto& s : sources) if (s.first == k) zip.add(codePath(stem, k), s.second);
    closeArchive(zip, stem, path);
}

}  // namespace pt
}  // namespace ppo
