/* parmfile.c -- host-side (C) HTK parameter files (HTKBook speechio.tex:835-950):
 *   12-byte big-endian header {int32 nSamples, int32 sampPeriod (100 ns), int16 sampSize (bytes), int16 parmKind}
 *   (ReadHTKHeader HWave.c:1408), rows of big-endian float32, or with _C (02000) int16 rows preceded by the float
 *   vectors A and B (4 "samples"), value = (short + B)/A (GetParm HParm.c:3488-3492); with _K (010000) a trailing
 *   16-bit checksum over all 16-bit words after the header, crc = (crc*65536 + word) % 36897 (UpdateCRCC HParm.c:3357).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

#define HASCOMPX 02000
#define HASCRCC  010000

static unsigned be16(const unsigned char *p) { return ((unsigned)p[0] << 8) | p[1]; }
static unsigned be32(const unsigned char *p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }
static float be_float(const unsigned char *p) { unsigned u = be32(p); float f; memcpy(&f, &u, 4); return f; }

void htkamd_free(void *p) { free(p); }

int htkamd_parm_read(const char *path, float **data, int *nFrames, int *nCols, int *sampPeriod, int *kind)
{
   FILE *f;
   unsigned char hdr[12], *buf = NULL;
   long n, fileBytes;
   int nSamples, sampSize, pk, cols, i, j, rc = HTKAMD_OK;
   size_t bodyBytes;
   float *out = NULL;
   if (!path || !data || !nFrames || !nCols) { htkamd_set_error("parm_read: NULL argument"); return HTKAMD_EINVAL; }
   f = fopen(path, "rb");
   if (!f) { htkamd_set_error("parm_read: cannot open %s", path); return HTKAMD_EINVAL; }
   if (fread(hdr, 1, 12, f) != 12) { fclose(f); htkamd_set_error("parm_read: %s: no HTK header", path); return HTKAMD_EINVAL; }
   nSamples = (int)be32(hdr); sampSize = (int)(short)be16(hdr + 8); pk = (int)be16(hdr + 10);
   if (sampPeriod) *sampPeriod = (int)be32(hdr + 4);
   fseek(f, 0, SEEK_END); fileBytes = ftell(f); fseek(f, 12, SEEK_SET);
   if (nSamples < 0 || sampSize <= 0) { fclose(f); htkamd_set_error("parm_read: %s: bad header", path); return HTKAMD_EINVAL; }
   if (sampSize % ((pk & HASCOMPX) ? 2 : 4)) {          /* a row is whole shorts (compressed) or whole floats */
      fclose(f); htkamd_set_error("parm_read: %s: sample size %d is not a multiple of %d", path, sampSize, (pk & HASCOMPX) ? 2 : 4); return HTKAMD_EINVAL;
   }
   bodyBytes = (size_t)nSamples * sampSize;
   if ((long)(12 + bodyBytes + ((pk & HASCRCC) ? 2 : 0)) > fileBytes) { fclose(f); htkamd_set_error("parm_read: %s: file shorter than its header says", path); return HTKAMD_EINVAL; }
   buf = (unsigned char *)malloc(bodyBytes + 2);          /* bounded by the file's own size (checked above) */
   if (!buf) { fclose(f); htkamd_set_error("parm_read: %s: out of memory (%zu bytes)", path, bodyBytes); return HTKAMD_ENOMEM; }
   n = (long)fread(buf, 1, bodyBytes + ((pk & HASCRCC) ? 2 : 0), f);
   fclose(f);
   if ((size_t)n != bodyBytes + ((pk & HASCRCC) ? 2 : 0)) { free(buf); htkamd_set_error("parm_read: %s: short read", path); return HTKAMD_EINVAL; }
   if (pk & HASCRCC) {
      unsigned crc = 0;
      size_t w;
      for (w = 0; w + 1 < bodyBytes + 1; w += 2) crc = (crc * 65536u + be16(buf + w)) % 36897u;
      if (crc != be16(buf + bodyBytes)) { free(buf); htkamd_set_error("parm_read: %s: CRC check failed", path); return HTKAMD_EINVAL; }
   }
   if (pk & HASCOMPX) {
      cols = sampSize / 2;
      if (nSamples < 4) { free(buf); htkamd_set_error("parm_read: %s: compressed file without A/B vectors", path); return HTKAMD_EINVAL; }
      nSamples -= 4;
      out = (float *)malloc(sizeof(float) * (size_t)(nSamples ? nSamples : 1) * cols);
      if (!out) { free(buf); htkamd_set_error("parm_read: %s: out of memory", path); return HTKAMD_ENOMEM; }
      for (i = 0; i < nSamples; i++)
         for (j = 0; j < cols; j++) {
            const float A = be_float(buf + 4 * j), B = be_float(buf + 4 * (cols + j));
            const short s = (short)be16(buf + 8 * cols + ((size_t)i * cols + j) * 2);
            out[(size_t)i * cols + j] = ((float)s + B) / A;
         }
   } else {
      cols = sampSize / 4;
      out = (float *)malloc(sizeof(float) * (size_t)(nSamples ? nSamples : 1) * cols);
      if (!out) { free(buf); htkamd_set_error("parm_read: %s: out of memory", path); return HTKAMD_ENOMEM; }
      { size_t z, nz = (size_t)nSamples * (size_t)cols; for (z = 0; z < nz; z++) out[z] = be_float(buf + 4 * z); }
   }
   free(buf);
   *data = out; *nFrames = nSamples; *nCols = cols;
   if (kind) *kind = pk & ~(HASCOMPX | HASCRCC);
   return rc;
}

int htkamd_parm_write(const char *path, const float *data, int nFrames, int nCols, int sampPeriod, int kind, int withCrc)
{
   FILE *f;
   unsigned char hdr[12], *buf;
   unsigned crc = 0;
   size_t i, n = (size_t)nFrames * nCols;
   int pk = (kind & ~(HASCOMPX | HASCRCC)) | (withCrc ? HASCRCC : 0), sz = nCols * 4;
   if (!path || (!data && n) || nFrames < 0 || nCols <= 0) { htkamd_set_error("parm_write: bad argument"); return HTKAMD_EINVAL; }
   f = fopen(path, "wb");
   if (!f) { htkamd_set_error("parm_write: cannot open %s", path); return HTKAMD_EINVAL; }
   hdr[0] = (unsigned char)(nFrames >> 24); hdr[1] = (unsigned char)(nFrames >> 16); hdr[2] = (unsigned char)(nFrames >> 8); hdr[3] = (unsigned char)nFrames;
   hdr[4] = (unsigned char)(sampPeriod >> 24); hdr[5] = (unsigned char)(sampPeriod >> 16); hdr[6] = (unsigned char)(sampPeriod >> 8); hdr[7] = (unsigned char)sampPeriod;
   hdr[8] = (unsigned char)(sz >> 8); hdr[9] = (unsigned char)sz; hdr[10] = (unsigned char)(pk >> 8); hdr[11] = (unsigned char)pk;
   fwrite(hdr, 1, 12, f);
   buf = (unsigned char *)malloc(4 * (n ? n : 1));
   if (!buf) { fclose(f); htkamd_set_error("parm_write: out of memory"); return HTKAMD_ENOMEM; }
   for (i = 0; i < n; i++) {
      unsigned u; memcpy(&u, data + i, 4);
      buf[4 * i] = (unsigned char)(u >> 24); buf[4 * i + 1] = (unsigned char)(u >> 16); buf[4 * i + 2] = (unsigned char)(u >> 8); buf[4 * i + 3] = (unsigned char)u;
      crc = (crc * 65536u + (u >> 16)) % 36897u;
      crc = (crc * 65536u + (u & 0xffffu)) % 36897u;
   }
   fwrite(buf, 1, 4 * n, f);
   if (withCrc) { unsigned char c[2] = {(unsigned char)(crc >> 8), (unsigned char)crc}; fwrite(c, 1, 2, f); }
   free(buf);
   fclose(f);
   return HTKAMD_OK;
}

/* ---- waveform files: the sources of the MFCC front end (SOURCEFORMAT = WAV | HTK) ------------------------------------
 * WAV : RIFF/WAVE little-endian, chunks walked until "data" as GetWAVHeaderInfo does (HWave.c:1052-1131); 16-bit PCM mono only
 *       (the reference also converts 8-bit, mu/a-law and stereo: rejected here), sampPeriod = 1e7 / rate in 100 ns units.
 * HTK : 12-byte big-endian header {nSamples, sampPeriod, sampSize = 2, kind = WAVEFORM (0)} + big-endian shorts
 *       (GetHTKHeaderInfo HWave.c:1408 ff.).
 * *samples is malloc'd (release with htkamd_free). */
static unsigned le32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
static unsigned le16(const unsigned char *p) { return p[0] | (p[1] << 8); }

int htkamd_wave_read(const char *path, int format, short **samples, long *nSamples, double *sampPeriod)
{
   if (!path || !samples || !nSamples) { htkamd_set_error("wave_read: NULL argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "rb");
   if (!f) { htkamd_set_error("wave_read: cannot open %s", path); return HTKAMD_EIO; }
   unsigned char h[16];
   short *out = NULL; long n = 0; double per = 0.0;
   if (format == HTKAMD_WAVE_HTK) {
      if (fread(h, 1, 12, f) != 12) { fclose(f); htkamd_set_error("wave_read: %s: short header", path); return HTKAMD_EINVAL; }
      const long ns = (long)((h[0] << 24) | (h[1] << 16) | (h[2] << 8) | h[3]);
      const long sp = (long)((h[4] << 24) | (h[5] << 16) | (h[6] << 8) | h[7]);
      const int size = (h[8] << 8) | h[9], kind = (h[10] << 8) | h[11];
      if ((kind & 077) != 0 || size != 2 || ns < 0) { fclose(f); htkamd_set_error("wave_read: %s is not an HTK WAVEFORM file (kind %o, sample size %d)", path, kind, size); return HTKAMD_EINVAL; }
      {  /* the header's count is bounded by what the file holds before anything is allocated */
         const long at = ftell(f); fseek(f, 0, SEEK_END); const long left = ftell(f) - at; fseek(f, at, SEEK_SET);
         if (ns > left / 2) { fclose(f); htkamd_set_error("wave_read: %s: file shorter than its header says", path); return HTKAMD_EINVAL; }
      }
      out = (short *)malloc(sizeof(short) * (size_t)(ns ? ns : 1));
      unsigned char *raw = (unsigned char *)malloc((size_t)(ns ? ns : 1) * 2);
      if (!out || !raw) { free(raw); free(out); fclose(f); htkamd_set_error("wave_read: %s: out of memory", path); return HTKAMD_ENOMEM; }
      if (fread(raw, 2, (size_t)ns, f) != (size_t)ns) { free(raw); free(out); fclose(f); htkamd_set_error("wave_read: %s: file shorter than its header says", path); return HTKAMD_EINVAL; }
      for (long i = 0; i < ns; i++) out[i] = (short)((raw[2 * i] << 8) | raw[2 * i + 1]);
      free(raw); n = ns; per = (double)sp;
   } else if (format == HTKAMD_WAVE_WAV) {
      if (fread(h, 1, 12, f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) { fclose(f); htkamd_set_error("wave_read: %s is not a RIFF/WAVE file", path); return HTKAMD_EINVAL; }
      int gotFmt = 0;
      for (;;) {
         if (fread(h, 1, 8, f) != 8) { fclose(f); htkamd_set_error("wave_read: %s: no data chunk", path); return HTKAMD_EINVAL; }
         unsigned len = le32(h + 4);
         if (!memcmp(h, "data", 4)) {
            if (!gotFmt) { fclose(f); htkamd_set_error("wave_read: %s: data chunk before fmt chunk", path); return HTKAMD_EINVAL; }
            {  /* a streamed WAV announces 0xFFFFFFFF (or 0): take what the file holds, as the reference's reader does for pipes */
               const long at = ftell(f); fseek(f, 0, SEEK_END); const long left = ftell(f) - at; fseek(f, at, SEEK_SET);
               if ((long)len > left || len == 0xFFFFFFFFu) len = (unsigned)left;
            }
            n = (long)(len / 2);
            out = (short *)malloc(sizeof(short) * (size_t)(n ? n : 1));
            unsigned char *raw = (unsigned char *)malloc((size_t)(n ? n : 1) * 2);
            if (!out || !raw) { free(raw); free(out); fclose(f); htkamd_set_error("wave_read: %s: out of memory", path); return HTKAMD_ENOMEM; }
            if (fread(raw, 2, (size_t)n, f) != (size_t)n) { free(raw); free(out); fclose(f); htkamd_set_error("wave_read: %s: data chunk shorter than its length field", path); return HTKAMD_EINVAL; }
            for (long i = 0; i < n; i++) out[i] = (short)le16(raw + 2 * i);
            free(raw);
            break;
         }
         if (!memcmp(h, "fmt ", 4)) {
            unsigned char fm[16];
            if (len < 16 || fread(fm, 1, 16, f) != 16) { fclose(f); htkamd_set_error("wave_read: %s: bad fmt chunk", path); return HTKAMD_EINVAL; }
            const unsigned type = le16(fm), chans = le16(fm + 2), rate = le32(fm + 4), bits = le16(fm + 14);
            if (type != 1 || chans != 1 || bits != 16 || rate == 0) {
               fclose(f); htkamd_set_error("wave_read: %s: only 16-bit mono PCM is supported (format %u, %u channels, %u bits)", path, type, chans, bits); return HTKAMD_EINVAL;
            }
            per = (double)(1.0E7f / (float)rate);                   /* w->sampPeriod = 1.0E7 / (float)lng (HWave.c:1107) */
            gotFmt = 1; len -= 16;
         }
         len += len & 1u;                                         /* RIFF chunks are padded to an even length */
         if (len && fseek(f, (long)len, SEEK_CUR)) { fclose(f); htkamd_set_error("wave_read: %s: truncated chunk", path); return HTKAMD_EINVAL; }
      }
   } else { fclose(f); htkamd_set_error("wave_read: unknown format %d", format); return HTKAMD_EINVAL; }
   fclose(f);
   *samples = out; *nSamples = n;
   if (sampPeriod) *sampPeriod = per;
   return HTKAMD_OK;
}
