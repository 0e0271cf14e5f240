// Error reporting, file and gzip helpers of the host front end.
#include "common.h"

#include <zlib.h>

#include <cstring>
#include <fstream>
#include <sys/stat.h>

namespace vgan {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

bool file_exists(const std::string &path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 && !S_ISDIR(st.st_mode);
}

bool gunzip_members(const void *data, size_t n, std::string &out) {
    const unsigned char *p = (const unsigned char *)data;
    out.clear();
    std::vector<unsigned char> buf(1 << 20);
    while (n > 0) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
        zs.next_in = const_cast<unsigned char *>(p);
        zs.avail_in = (uInt)std::min<size_t>(n, 0x7fffffffu);
        int rc;
        do {
            zs.next_out = buf.data();
            zs.avail_out = (uInt)buf.size();
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&zs);
                return false;
            }
            out.append((const char *)buf.data(), buf.size() - zs.avail_out);
            if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) { // truncated member
                inflateEnd(&zs);
                return false;
            }
        } while (rc != Z_STREAM_END);
        const size_t used = (size_t)(zs.next_in - p);
        inflateEnd(&zs);
        p += used;
        n -= used;
    }
    return true;
}

bool gzip_bytes(const std::string &in, std::string &out) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 1, Z_DEFLATED, 16 + MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, in.size()) + 32);
    zs.next_in = (unsigned char *)in.data();
    zs.avail_in = (uInt)in.size();
    zs.next_out = (unsigned char *)&out[0];
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return rc == Z_STREAM_END;
}

bool read_file(const std::string &path, std::string &out, bool inflate_if_gzip) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (inflate_if_gzip && raw.size() >= 2 && (unsigned char)raw[0] == 0x1f && (unsigned char)raw[1] == 0x8b) {
        return gunzip_members(raw.data(), raw.size(), out);
    }
    out.swap(raw);
    return true;
}

bool write_file(const std::string &path, const std::string &bytes) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write(bytes.data(), (std::streamsize)bytes.size());
    return (bool)f;
}

bool read_text_maybe_gz(const std::string &path, std::string &out) {
    if (file_exists(path)) return read_file(path, out);
    if (file_exists(path + ".gz")) return read_file(path + ".gz", out);
    return false;
}

} // namespace vgan

extern "C" const char *vgan_last_error(void) { return vgan::last_error(); }
extern "C" int vgan_abi_version(void) { return VGAN_ABI_VERSION; }
