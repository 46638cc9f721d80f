/* mmf.c -- HTK model definition files (text MMF / one-HMM-per-file) <-> the flat htkamd_model_desc.
 *
 * Replaces, for the model kinds the hot path supports (diagonal covariance, continuous densities, PLAINHS/SHAREDHS; one stream,
 * or several -- <STREAMINFO> S, <NUMMIXES> per stream, <SWEIGHTS>, <STREAM> s, ~v varFloorN: every vector of a stream is held at the
 * stream's dimensions of an undivided row, htkamd_host_stream_dims --): LoadHMMSet (HModel.c:3809) = MakeHMMSet (:3580, the HMM list "logical [physical]") +
 * LoadAllMacros / LoadMacroFiles (:3721) + the -d directory search; the grammar of GetOptions (:1649),
 * GetMean/GetVariance/GetTransMat (:1737-1990: transitions become log values, <= MINLARG -> LZERO), GetMixPDF (:2140),
 * GetStateInfo (:2350: a missing <MIXTURE> index is a pruned component of weight 0), GetHMMDef (:2490); and the
 * writer SaveHMMSet (:4979) = SaveMacros (:4342: ~o options, then ~t, ~s, ~h macros in hash-table order) with
 * PutStateInfo (:3053), PutMixPDF (:3029), PutTransMat (:2877: rows renormalised in float) and WriteFloat's " %e".
 * Binary definitions (':' + code byte keywords, big-endian numbers: PutSymbol :2581, Token.binForm :505) are read and
 * written as well, shared mixture pdfs (~m) are kept shared (one Gaussian, several components).  Shared mean / variance vectors
 * (~u / ~v macros referenced inside a mixture, GetMean :1737 / GetVariance :1770) are read, kept (every Gaussian holds its copy of the
 * values plus the number of the macro it shares: htkamd_mmf_sharing) and written back as macros; duration vectors (<DURATION>, ~d macros; the set's <POISSOND> / <GAMMAD> / <GEND> kind) are carried and written
 * back (round 6); ~u / ~v macros inside multi-stream sets and
 * transforms are rejected with HTKAMD_EMODEL: they do not occur on the path's configurations.
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

typedef struct { char *name; int nMix; int comp0; int inlineOwner; int src; int *sMix; float *sw; int dur; } mmf_state;   /* name NULL = un-named (inline); src: index of the file it came from;
   several streams: nMix = all components of the state, sMix[s] those of stream s (consecutive from comp0), sw = <SWEIGHTS> or NULL; dur: its duration vector (dm[]) or -1 */
typedef struct { char *name; int N; int off; int src; } mmf_trans;
typedef struct { char *name; int N; int *state; int trans; int src; int dur; } mmf_hmm;     /* src: index of the file it came from; dur: the model's duration vector (dm[]) or -1 */

struct htkamd_mmf {
   int vecSize, streamWidth, hasOpts;
   int nStreams, swidth[8], *dimStream, curStream;                  /* several streams (<STREAMINFO> S w1..wS): dimension -> stream over the undivided vector */
   char kind[64], cov[16], dur[16], setId[128];
   /* pools */
   mmf_state *st; int nSt, capSt;
   float *wt; int *cg; int nComp, capComp;
   float *mean, *var, *gconst; unsigned char *hasG; int nG, capG;
   char **gName; int *gSrc; int capGN;                              /* ~m macro name of Gaussian g or NULL, and the file it was defined in */
   mmf_trans *tr; int nTr, capTr; float *tp; int nTp, capTp;       /* tp: LOG transition values */
   mmf_hmm *hm; int nHm, capHm;
   float *varFloor;                                                 /* ~v "varFloor1" or NULL */
   /* shared vectors: ~u (means) and ~v (variances) macros; gMeanMac/gVarMac[g] = macro of Gaussian g's mean / variance or -1 */
   struct { char type; char *name; float *v; int src; int stream; } *vm; int nVm, capVm;      /* stream: of a varFloorN macro of a multi-stream set, else -1 */
   int *gMeanMac, *gVarMac; int capMac;
   /* ~w stream-weight macros (GetSWeights HModel.c:1621): nStreams numbers under a name; a state that names one takes a copy (the set is written
      back with <SWEIGHTS> in the states) */
   struct { char *name; float w[8]; } *wm; int nWm, capWm;
   /* duration vectors (GetDuration HModel.c:1580, PutDuration :2840): <DURATION> n v1..vn behind a state's streams or a model's transition matrix, inline or as
      a ~d macro.  Nothing on the path reads them (HERest / HVite neither): they are carried and written back where they stood */
   struct { char *name; int n; float *v; int src; } *dm; int nDm, capDm;
   /* logical list */
   char **logName; int *logPhys; int nLog; int *logSorted;        /* logSorted: list positions in name order (stable) */
   /* desc arrays */
   htkamd_model_desc d; int *stateCompOff, *transN, *transOff, *hmmTrans, *hmmStateOff, *hmmState;
   int *gStr;                                                       /* stream of Gaussian g */
   float *swAll;                                                    /* [nSt*NS] stream weights for the desc */
   int tiedMix;                                                     /* hsKind TIEDHS: <TMIX> streams (GetStream HModel.c:1878-1892) */
   char *tmName[8]; int tmM[8];                                     /* per stream: generic ~m macro name and pool size (tmRecs[s].mixId / nMix) */
   int *gPend; int capPend, lastVecN;                                         /* a ~m macro of a multi-stream set read before its stream is known: its width, values at [0..width) of the row */
   int finished, nFiles;
};

/* ------------------------------------------------------------------------------------------ tokenizer */
typedef struct { FILE *f; const char *path; int line; int pushed; char tok[256]; int kind; int bin; } rd;   /* bin: the last keyword was a binary symbol, so its numbers are binary too (Token.binForm, HModel.c:505,570) */
enum { T_EOF, T_MACRO, T_KEY, T_WORD };

static int rd_getc(rd *r) { int c = getc_unlocked(r->f); if (c == '\n') r->line++; return c; }    /* one reader per FILE: no lock per character */
static void rd_ungetc(rd *r, int c) { if (c == EOF) return; if (c == '\n') r->line--; ungetc(c, r->f); }

static int rd_next(rd *r)
{
   int c, n = 0;
   if (r->pushed) { r->pushed = 0; return r->kind; }
   do c = rd_getc(r); while (c != EOF && isspace(c));
   if (c == EOF) return r->kind = T_EOF;
   if (c == '~') {
      c = rd_getc(r);
      r->tok[0] = (char)tolower(c); r->tok[1] = 0;
      return r->kind = T_MACRO;
   }
   if (c == '<') {
      while ((c = rd_getc(r)) != EOF && c != '>' && n < 250) r->tok[n++] = (char)toupper(c);
      r->tok[n] = 0;
      r->bin = 0;
      return r->kind = T_KEY;
   }
   if (c == '"' || c == '\'') {
      const int q = c;
      while ((c = rd_getc(r)) != EOF && c != q && n < 250) {
         if (c == '\\') c = rd_getc(r);
         r->tok[n++] = (char)c;
      }
      r->tok[n] = 0;
      return r->kind = T_WORD;
   }
   if (c == ':') {                                   /* binary symbol: ':' + code byte (PutSymbol HModel.c:2581, enum Symbol :394) */
      static const char *const code[32] = {"BEGINHMM", "USE", "ENDHMM", "NUMMIXES", "NUMSTATES", "STREAMINFO", "VECSIZE", "NULLD", "POISSOND", "GAMMAD",
         "RELD", "GEND", "DIAGC", "FULLC", "XFORMC", "STATE", "TMIX", "MIXTURE", "STREAM", "SWEIGHTS", "MEAN", "VARIANCE", "INVCOVAR", "XFORM", "GCONST",
         "DURATION", "INVDIAGC", "TRANSP", "DPROB", "LLTC", "LLTCOVAR", "PROJSIZE"};
      c = rd_getc(r);
      if (c >= 0 && c < 32) snprintf(r->tok, sizeof(r->tok), "%s", code[c]);
      else if (c == 110) snprintf(r->tok, sizeof(r->tok), "RCLASS");
      else if (c == 119) snprintf(r->tok, sizeof(r->tok), "HMMSETID");
      else snprintf(r->tok, sizeof(r->tok), "?BINARY%d", c);
      r->bin = 1;
      return r->kind = T_KEY;
   }
   do {
      if (c == '\\') c = rd_getc(r);
      r->tok[n++] = (char)c;
      c = rd_getc(r);
   } while (c != EOF && !isspace(c) && c != '<' && c != '~' && n < 250);
   rd_ungetc(r, c);
   r->tok[n] = 0;
   return r->kind = T_WORD;
}
static void rd_push(rd *r) { r->pushed = 1; }

static int fail(rd *r, const char *what)
{
   htkamd_set_error("%s:%d: %s (at '%s')", r->path, r->line, what, r->tok);
   return HTKAMD_EMODEL;
}
static int rd_int(rd *r, int *v)
{
   char *e;
   if (r->bin && !r->pushed) {                       /* ReadShort, big-endian (HShell.c:1545) */
      const int hi = getc_unlocked(r->f), lo = getc_unlocked(r->f);
      if (lo == EOF) return fail(r, "unexpected end of binary data");
      *v = (short)((hi << 8) | lo);
      return HTKAMD_OK;
   }
   if (rd_next(r) != T_WORD) return fail(r, "integer expected");
   *v = (int)strtol(r->tok, &e, 10);
   return *e ? fail(r, "integer expected") : HTKAMD_OK;
}
/* Decimal text -> the correctly rounded float without strtof for the numbers model files are made of ([-]d.dddddde[+-]dd, up to 15
   digits): digits as an integer m < 2^53, then m * 10^k or m / 10^k with k <= 22 -- both operands exact doubles, so the double is
   correctly rounded (Clinger's fast path) -- and the double rounded to float.  The second rounding can only go wrong when the double
   lies within its own rounding error of the midpoint of two floats: those (and everything unusual) are left to strtof.  Returns 1 if
   *v was set. */
static int fast_float(const char *s, float *v)
{
   static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
   unsigned long long m = 0;
   int neg = 0, nd = 0, frac = 0, e10 = 0, seen = 0;
   if (*s == '-') { neg = 1; s++; } else if (*s == '+') s++;
   for (; *s >= '0' && *s <= '9'; s++) { seen = 1; if (nd < 15) { m = m * 10 + (unsigned)(*s - '0'); if (m) nd++; } else return 0; }
   if (*s == '.') for (s++; *s >= '0' && *s <= '9'; s++) { seen = 1; if (nd < 15) { m = m * 10 + (unsigned)(*s - '0'); if (m) nd++; frac++; } else return 0; }
   if (!seen) return 0;
   if (*s == 'e' || *s == 'E') {
      int es = 1, ev = 0, ed = 0;
      s++;
      if (*s == '-') { es = -1; s++; } else if (*s == '+') s++;
      for (; *s >= '0' && *s <= '9'; s++) { ev = ev * 10 + (*s - '0'); if (++ed > 4) return 0; }
      if (!ed) return 0;
      e10 = es * ev;
   }
   if (*s) return 0;
   e10 -= frac;
   if (m == 0) { *v = neg ? -0.0f : 0.0f; return 1; }
   if (e10 < -22 || e10 > 22) return 0;
   const double d = (e10 < 0) ? (double)m / p10[-e10] : (double)m * p10[e10];
   unsigned long long u;
   memcpy(&u, &d, 8);
   const unsigned int low = (unsigned int)(u & 0x1FFFFFFFu);            /* the 29 mantissa bits a float does not keep */
   if (low >= 0x0FFFFFFEu && low <= 0x10000002u) return 0;               /* at a float midpoint to within the double's own error */
   const float f = (float)d;
   if (!(f > 1.0e-30f || f < -1.0e-30f) || f > 1.0e30f || f < -1.0e30f) return 0;   /* near the float range's ends: strtof */
   *v = neg ? -f : f;
   return 1;
}
static int rd_float(rd *r, float *v)
{
   char *e;
   if (r->bin && !r->pushed) {                       /* ReadFloat, big-endian IEEE (HShell.c:1600) */
      unsigned char b[4];
      const int c0 = getc_unlocked(r->f), c1 = getc_unlocked(r->f), c2 = getc_unlocked(r->f), c3 = getc_unlocked(r->f);
      if (c3 == EOF) return fail(r, "unexpected end of binary data");
      b[0] = (unsigned char)c0; b[1] = (unsigned char)c1; b[2] = (unsigned char)c2; b[3] = (unsigned char)c3;
      const unsigned int u = ((unsigned int)b[0] << 24) | ((unsigned int)b[1] << 16) | ((unsigned int)b[2] << 8) | b[3];
      memcpy(v, &u, 4);
      return HTKAMD_OK;
   }
   if (rd_next(r) != T_WORD) return fail(r, "number expected");
   if (fast_float(r->tok, v)) return HTKAMD_OK;
   *v = strtof(r->tok, &e);                         /* fscanf("%e") into a float, HShell.c ReadFloat */
   return *e ? fail(r, "number expected") : HTKAMD_OK;
}
static int rd_name(rd *r, char **out)
{
   if (rd_next(r) != T_WORD) return fail(r, "macro name expected");
   *out = strdup(r->tok);
   return HTKAMD_OK;
}

#define GROW(p, n, cap, need, T) do { if ((n) + (need) > (cap)) { (cap) = ((n) + (need)) * 2 + 16; (p) = (T *)realloc((p), sizeof(T) * (size_t)(cap)); } } while (0)

static int is_parm_kind(const char *t)
{
   static const char *base[] = {"WAVEFORM", "LPC", "LPREFC", "LPCEPSTRA", "LPDELCEP", "IREFC", "MFCC", "FBANK", "MELSPEC", "USER", "DISCRETE", "PLP", NULL};
   for (int i = 0; base[i]; i++) {
      size_t n = strlen(base[i]);
      if (!strncmp(t, base[i], n) && (t[n] == 0 || t[n] == '_')) return 1;
   }
   return 0;
}

/* ------------------------------------------------------------------------------------------ grammar */
static int parse_options(struct htkamd_mmf *s, rd *r)
{
   int rc;
   for (;;) {
      int k = rd_next(r);
      if (k != T_KEY) { rd_push(r); return HTKAMD_OK; }
      const char *t = r->tok;
      if (!strcmp(t, "STREAMINFO")) {
         int S, w;
         if ((rc = rd_int(r, &S))) return rc;
         if (S < 1 || S > 7) return fail(r, "<STREAMINFO>: 1..7 streams");
         if (s->nStreams && s->nStreams != S) return fail(r, "inconsistent number of streams");
         for (int i = 0; i < S; i++) {
            if ((rc = rd_int(r, &w))) return rc;
            if (s->nStreams && s->swidth[i] != w) return fail(r, "inconsistent stream width");
            s->swidth[i] = w;
         }
         s->nStreams = S;
         if (S == 1) s->streamWidth = s->swidth[0];
      } else if (!strcmp(t, "VECSIZE")) {
         int v;
         if ((rc = rd_int(r, &v))) return rc;
         if (s->vecSize && s->vecSize != v) return fail(r, "inconsistent vector size");
         s->vecSize = v;
      } else if (!strcmp(t, "HMMSETID")) {
         if (rd_next(r) != T_WORD) return fail(r, "set id expected");
         snprintf(s->setId, sizeof(s->setId), "%.127s", r->tok);
      } else if (!strcmp(t, "NULLD") || !strcmp(t, "POISSOND") || !strcmp(t, "GAMMAD") || !strcmp(t, "GEND") || !strcmp(t, "RELD")) {
         if (s->dur[0] && strcmp(s->dur, t)) return fail(r, "inconsistent duration kind");
         snprintf(s->dur, sizeof(s->dur), "%.15s", t);                        /* (carried: no tool on the path evaluates a duration model) */
      }
      else if (!strcmp(t, "DIAGC")) snprintf(s->cov, sizeof(s->cov), "%.15s", t);
      else if (!strcmp(t, "FULLC") || !strcmp(t, "XFORMC") || !strcmp(t, "LLTC") || !strcmp(t, "INVDIAGC")) return fail(r, "only DIAGC covariances are supported");
      else if (is_parm_kind(t)) {
         if (s->kind[0] && strcmp(s->kind, t)) return fail(r, "inconsistent parameter kind");
         snprintf(s->kind, sizeof(s->kind), "%.63s", t);
      } else if (!strcmp(t, "PROJSIZE") || !strcmp(t, "INPUTXFORM") || !strcmp(t, "PARENTXFORM") || !strcmp(t, "MSDINFO") ||
                 !strcmp(t, "DISCRETE") || !strcmp(t, "DPROB")) {
         return fail(r, "unsupported global option");
      } else { rd_push(r); return HTKAMD_OK; }      /* structural keyword: the caller's business */
      s->hasOpts = 1;
   }
}

/* several streams: the map dimension -> stream, once the kind and the widths are known */
static int need_stream_dims(struct htkamd_mmf *s, rd *r)
{
   char why[160];
   if (s->dimStream) return HTKAMD_OK;
   if (s->vecSize == 0) return fail(r, "<VECSIZE> must precede the first vector");
   s->dimStream = (int *)malloc(sizeof(int) * (size_t)s->vecSize);
   if (htkamd_host_stream_dims(s->kind, s->vecSize, s->nStreams, s->swidth, s->dimStream, why, sizeof(why))) {
      free(s->dimStream); s->dimStream = NULL;
      htkamd_set_error("%s:%d: <STREAMINFO>: %s", r->path, r->line, why);
      return HTKAMD_EMODEL;
   }
   return HTKAMD_OK;
}

/* A vector of the whole observation, or (several streams) of stream `stream`: its values go to the stream's dimensions of the
   undivided row dst, the other dimensions are set to `fill` (0 for means; +infinity for variances: such a dimension contributes
   x*x/inf = 0 and 1/variance = 0 to every score). */
static int parse_vector_ms(struct htkamd_mmf *s, rd *r, float *dst, int stream, float fill)
{
   int n, rc;
   if ((rc = rd_int(r, &n))) return rc;
   if (s->nStreams > 1 && stream == -1) {               /* a ~m macro defined outside any state: its stream is known when it is first used */
      if (s->vecSize == 0) return fail(r, "<VECSIZE> must precede the first vector");
      if (n < 1 || n > s->vecSize) return fail(r, "vector size");
      for (int i = 0; i < s->vecSize; i++) { if (i >= n) { dst[i] = fill; continue; } if ((rc = rd_float(r, dst + i))) return rc; }
      s->lastVecN = n;
      return HTKAMD_OK;
   }
   if (s->nStreams > 1) {
      if ((rc = need_stream_dims(s, r))) return rc;
      if (stream < 0 || stream >= s->nStreams) return fail(r, "vector outside a stream");
      if (n != s->swidth[stream]) return fail(r, "vector size differs from the stream's width");
      for (int i = 0; i < s->vecSize; i++) {
         if (s->dimStream[i] != stream) { dst[i] = fill; continue; }
         if ((rc = rd_float(r, dst + i))) return rc;
      }
      return HTKAMD_OK;
   }
   if (s->vecSize == 0) s->vecSize = n;
   if (n != s->vecSize) return fail(r, "vector size differs from <VECSIZE>");
   for (int i = 0; i < n; i++) if ((rc = rd_float(r, dst + i))) return rc;
   return HTKAMD_OK;
}

static void gname_set(struct htkamd_mmf *s, int g, char *name)
{
   if (g + 1 > s->capGN) {
      const int nc = (g + 1) * 2 + 16;
      s->gName = (char **)realloc(s->gName, sizeof(char *) * (size_t)nc);
      s->gSrc = (int *)realloc(s->gSrc, sizeof(int) * (size_t)nc);
      for (int i = s->capGN; i < nc; i++) { s->gName[i] = NULL; s->gSrc[i] = 0; }
      s->capGN = nc;
   }
   s->gName[g] = name; s->gSrc[g] = s->nFiles;
}
static int find_gauss(const struct htkamd_mmf *s, const char *name)
{
   for (int g = 0; g < s->nG && g < s->capGN; g++) if (s->gName[g] && !strcmp(s->gName[g], name)) return g;
   return -1;
}

static int find_vmacro(const struct htkamd_mmf *s, char type, const char *name)
{
   for (int i = 0; i < s->nVm; i++) if (s->vm[i].type == type && !strcmp(s->vm[i].name, name)) return i;
   return -1;
}
static void mac_set(struct htkamd_mmf *s, int g, int meanMac, int varMac)
{
   if (g + 1 > s->capMac) {
      const int nc = (g + 1) * 2 + 16;
      s->gMeanMac = (int *)realloc(s->gMeanMac, sizeof(int) * (size_t)nc);
      s->gVarMac = (int *)realloc(s->gVarMac, sizeof(int) * (size_t)nc);
      for (int i = s->capMac; i < nc; i++) { s->gMeanMac[i] = -1; s->gVarMac[i] = -1; }
      s->capMac = nc;
   }
   s->gMeanMac[g] = meanMac; s->gVarMac[g] = varMac;
}
/* "<MEAN> n v.." / "<VARIANCE> n v.." inline, or a reference "~u name" / "~v name" to a macro defined earlier */
static int parse_shared_vector(struct htkamd_mmf *s, rd *r, int k, char type, const char *key, float *dst, int *mac)
{
   int rc;
   *mac = -1;
   if (k == T_MACRO && r->tok[0] == type) {
      char *nm;
      if (s->nStreams > 1) return fail(r, "~u / ~v macros in a multi-stream set are not supported");
      if ((rc = rd_name(r, &nm))) return rc;
      const int i = find_vmacro(s, type, nm);
      if (i < 0) { rc = fail(r, type == 'u' ? "undefined ~u macro" : "undefined ~v macro"); free(nm); return rc; }
      free(nm);
      memcpy(dst, s->vm[i].v, sizeof(float) * (size_t)s->vecSize);
      *mac = i;
      return HTKAMD_OK;
   }
   if (k == T_MACRO) return fail(r, "unexpected macro inside a mixture");
   if (k != T_KEY || strcmp(r->tok, key)) return fail(r, type == 'u' ? "<MEAN> expected" : "<VARIANCE> expected (DIAGC only)");
   return parse_vector_ms(s, r, dst, s->curStream, type == 'u' ? 0.0f : INFINITY);
}

static void pend_set(struct htkamd_mmf *s, int g, int v)
{
   if (g + 1 > s->capPend) {
      const int nc = (g + 1) * 2 + 16;
      s->gPend = (int *)realloc(s->gPend, sizeof(int) * (size_t)nc);
      for (int i = s->capPend; i < nc; i++) s->gPend[i] = 0;
      s->capPend = nc;
   }
   s->gPend[g] = v;
}
/* A ~m macro of a multi-stream set is used by stream `stream`: its vectors move to that stream's dimensions of the undivided row (first
   use), or it must have been used by the same stream before. */
static int resolve_stream(struct htkamd_mmf *s, rd *r, int g, int stream)
{
   if (s->nStreams <= 1) return HTKAMD_OK;
   const int pend = (g < s->capPend) ? s->gPend[g] : 0;
   if (pend < 0) return (-pend - 1 == stream) ? HTKAMD_OK : fail(r, "a ~m macro is used by two streams");
   if (pend == 0) return fail(r, "a ~m macro defined inside a stream cannot be shared here");
   int rc;
   if ((rc = need_stream_dims(s, r))) return rc;
   if (pend != s->swidth[stream]) return fail(r, "~m macro: vector size differs from the stream's width");
   const int D = s->vecSize;
   float *tmp = (float *)malloc(sizeof(float) * (size_t)pend);
   for (int pass = 0; pass < 2; pass++) {
      float *row = (pass ? s->var : s->mean) + (size_t)g * D;
      memcpy(tmp, row, sizeof(float) * (size_t)pend);
      int j = 0;
      for (int i = 0; i < D; i++) row[i] = (s->dimStream[i] == stream) ? tmp[j++] : (pass ? INFINITY : 0.0f);
   }
   free(tmp);
   pend_set(s, g, -(stream + 1));
   return HTKAMD_OK;
}

static int parse_mixpdf(struct htkamd_mmf *s, rd *r, int *gOut)
{
   int rc, k = rd_next(r);
   if (k == T_MACRO && r->tok[0] == 'm') {               /* reference to a shared mixture pdf */
      char *nm;
      if ((rc = rd_name(r, &nm))) return rc;
      const int g = find_gauss(s, nm);
      if (g < 0) { rc = fail(r, "undefined ~m macro"); free(nm); return rc; }
      free(nm);
      if ((rc = resolve_stream(s, r, g, s->curStream))) return rc;
      *gOut = g;
      return HTKAMD_OK;
   }
   if (k == T_KEY && !strcmp(r->tok, "RCLASS")) { int x; if ((rc = rd_int(r, &x))) return rc; k = rd_next(r); }
   if (s->vecSize == 0) return fail(r, "<VECSIZE> must precede the first mean");
   GROW(s->gconst, s->nG, s->capG, 1, float);
   { const int cap = s->capG;
     s->mean = (float *)realloc(s->mean, sizeof(float) * (size_t)cap * s->vecSize);
     s->var = (float *)realloc(s->var, sizeof(float) * (size_t)cap * s->vecSize);
     s->hasG = (unsigned char *)realloc(s->hasG, (size_t)cap); }
   const int g = s->nG;
   int meanMac, varMac;
   if ((rc = parse_shared_vector(s, r, k, 'u', "MEAN", s->mean + (size_t)g * s->vecSize, &meanMac))) return rc;
   k = rd_next(r);
   if ((rc = parse_shared_vector(s, r, k, 'v', "VARIANCE", s->var + (size_t)g * s->vecSize, &varMac))) return rc;
   if (meanMac >= 0 || varMac >= 0) mac_set(s, g, meanMac, varMac);
   s->hasG[g] = 0; s->gconst[g] = 0.0f;
   k = rd_next(r);
   if (k == T_KEY && !strcmp(r->tok, "GCONST")) { if ((rc = rd_float(r, s->gconst + g))) return rc; s->hasG[g] = 1; }
   else rd_push(r);
   s->nG++;
   *gOut = g;
   return HTKAMD_OK;
}

/* state body after "~s name" or "<STATE> i": returns the new state index (GetStateInfo HModel.c:1924, GetStream :1850) */
/* <DURATION> n v1 .. vn (the keyword has been read) -> a new entry of dm[] */
static int parse_duration_body(struct htkamd_mmf *s, rd *r, char *name, int *dOut)
{
   int n, rc;
   if ((rc = rd_int(r, &n))) return rc;
   if (n < 1 || n > 4096) return fail(r, "bad size of a duration vector");
   float *v = (float *)malloc(sizeof(float) * (size_t)n);
   for (int i = 0; i < n; i++) if ((rc = rd_float(r, v + i))) { free(v); return rc; }
   GROW(s->dm, s->nDm, s->capDm, 1, __typeof__(*s->dm));
   s->dm[s->nDm].name = name; s->dm[s->nDm].n = n; s->dm[s->nDm].v = v; s->dm[s->nDm].src = s->nFiles;
   *dOut = s->nDm++;
   return HTKAMD_OK;
}
/* where a duration may stand (behind a state's streams, behind a model's transition matrix): <DURATION> ..., a ~d reference, or nothing (*dOut = -1, token pushed back) */
static int parse_duration_opt(struct htkamd_mmf *s, rd *r, int *dOut)
{
   const int k = rd_next(r);
   *dOut = -1;
   if (k == T_KEY && !strcmp(r->tok, "DURATION")) return parse_duration_body(s, r, NULL, dOut);
   if (k == T_MACRO && r->tok[0] == 'd') {
      char *nm;
      int rc;
      if ((rc = rd_name(r, &nm))) return rc;
      for (int q = 0; q < s->nDm; q++) if (s->dm[q].name && !strcmp(s->dm[q].name, nm)) *dOut = q;
      free(nm);
      if (*dOut < 0) return fail(r, "undefined ~d macro");
      return HTKAMD_OK;
   }
   rd_push(r);
   return HTKAMD_OK;
}

static int parse_state_body(struct htkamd_mmf *s, rd *r, char *name, int *sOut)
{
   const int S = s->nStreams > 1 ? s->nStreams : 1;
   int rc, M = 0, k = rd_next(r), nMixS[8], off[9], seen[8] = {0};
   float *sw = NULL;
   for (int i = 0; i < S; i++) nMixS[i] = 1;
   if (k == T_KEY && !strcmp(r->tok, "NUMMIXES")) { for (int i = 0; i < S; i++) if ((rc = rd_int(r, &nMixS[i]))) return rc; k = rd_next(r); }
   off[0] = 0;
   for (int i = 0; i < S; i++) { if (nMixS[i] < 1) return fail(r, "bad <NUMMIXES>"); off[i + 1] = off[i] + nMixS[i]; }
   M = off[S];
   if (k == T_MACRO && r->tok[0] == 'w') {               /* a reference to a ~w macro where <SWEIGHTS> would stand (GetStateInfo HModel.c:1966) */
      char *nm;
      if ((rc = rd_name(r, &nm))) return rc;
      int i = -1;
      for (int q = 0; q < s->nWm; q++) if (!strcmp(s->wm[q].name, nm)) { i = q; break; }
      free(nm);
      if (i < 0) return fail(r, "undefined ~w macro");
      sw = (float *)malloc(sizeof(float) * (size_t)S);
      memcpy(sw, s->wm[i].w, sizeof(float) * (size_t)S);
      k = rd_next(r);
   }
   else if (k == T_KEY && !strcmp(r->tok, "SWEIGHTS")) {
      int n;
      if ((rc = rd_int(r, &n))) return rc;
      if (n != S) return fail(r, "incorrect number of stream weights");
      sw = (float *)malloc(sizeof(float) * (size_t)S);
      for (int i = 0; i < S; i++) if ((rc = rd_float(r, sw + i))) { free(sw); return rc; }
      k = rd_next(r);
   }
   GROW(s->st, s->nSt, s->capSt, 1, mmf_state);
   GROW(s->wt, s->nComp, s->capComp, M, float);
   s->cg = (int *)realloc(s->cg, sizeof(int) * (size_t)s->capComp);
   const int c0 = s->nComp;
   for (int m = 0; m < M; m++) { s->wt[c0 + m] = 0.0f; s->cg[c0 + m] = -1; }
   for (;;) {
      int stream = 0;
      if (k == T_KEY && !strcmp(r->tok, "STREAM")) {
         int x;
         if ((rc = rd_int(r, &x))) { free(sw); return rc; }
         if (x < 1 || x > S) { free(sw); return fail(r, "stream index out of range"); }
         stream = x - 1;
         k = rd_next(r);
      }
      if (seen[stream]) { free(sw); return fail(r, "stream defined twice"); }
      seen[stream] = 1;
      s->curStream = stream;
      const int cs = c0 + off[stream], Ms = nMixS[stream];
      if (k == T_KEY && !strcmp(r->tok, "DPROB")) { free(sw); return fail(r, "discrete streams are not supported"); }
      if (k == T_KEY && !strcmp(r->tok, "TMIX")) {
         /* GetTiedMixtures (HModel.c:991): the generic macro name; the pool is ~m "name1" .. "nameM"; then the compact weights */
         if (rd_next(r) != T_WORD) { free(sw); return fail(r, "tied mix macro name expected"); }
         if (!s->tmName[stream]) {
            s->tmName[stream] = strdup(r->tok); s->tmM[stream] = Ms;
         } else if (strcmp(s->tmName[stream], r->tok)) { free(sw); return fail(r, "bad generic ~m macro name in <TMIX>"); }
         else if (s->tmM[stream] != Ms) { free(sw); return fail(r, "inconsistent number of mixtures in <TMIX>"); }
         s->tiedMix = 1;
         for (int m = 0; m < Ms; m++) {
            char nm[320];
            snprintf(nm, sizeof(nm), "%s%d", s->tmName[stream], m + 1);
            const int g = find_gauss(s, nm);
            if (g < 0) { free(sw); return fail(r, "unknown tied mix macro (~m name<m>)"); }
            if ((rc = resolve_stream(s, r, g, stream))) { free(sw); return rc; }
            s->cg[cs + m] = g;
         }
         {  /* GetTiedWeights (HModel.c:882): a weight may be followed by *n (text) or come as weight - 2 with a count byte (binary) */
            float wv = 0.0f; int rep = 0;
            for (int m = 0; m < Ms; m++) {
               if (rep > 0) --rep;
               else if (r->bin) {
                  if ((rc = rd_float(r, &wv))) { free(sw); return rc; }
                  if (wv < 0.0f) { const int cnt = getc_unlocked(r->f); if (cnt == EOF) { free(sw); return fail(r, "repeat count expected"); } rep = cnt - 1; wv = wv + 2.0f; }
               } else {
                  if (rd_next(r) != T_WORD) { free(sw); return fail(r, "tied weight expected"); }
                  char *star = strchr(r->tok, '*'), *e;
                  if (star) { *star = 0; rep = (int)strtol(star + 1, &e, 10) - 1; if (*e || rep < 0) { free(sw); return fail(r, "repeat count expected"); } }
                  if (!fast_float(r->tok, &wv)) { wv = strtof(r->tok, &e); if (*e) { free(sw); return fail(r, "tied weight expected"); } }
               }
               s->wt[cs + m] = wv;
            }
         }
         k = rd_next(r);
         if (!(k == T_KEY && !strcmp(r->tok, "STREAM"))) { rd_push(r); break; }
         continue;
      }
      if (s->tiedMix) { free(sw); return fail(r, "a tied-mixture set (<TMIX>) cannot hold ordinary mixtures"); }
      if (k == T_KEY && !strcmp(r->tok, "MIXTURE")) {
         while (k == T_KEY && !strcmp(r->tok, "MIXTURE")) {
            int m; float w;
            if ((rc = rd_int(r, &m)) || (rc = rd_float(r, &w))) { free(sw); return rc; }
            if (m < 1 || m > Ms) { free(sw); return fail(r, "mixture index out of range"); }
            if (s->cg[cs + m - 1] >= 0) { free(sw); return fail(r, "mixture defined twice"); }
            s->wt[cs + m - 1] = w;
            if ((rc = parse_mixpdf(s, r, &s->cg[cs + m - 1]))) { free(sw); return rc; }
            k = rd_next(r);
         }
         rd_push(r);
      } else {
         if (Ms != 1) { free(sw); return fail(r, "<MIXTURE> expected"); }
         rd_push(r);
         s->wt[cs] = 1.0f;
         if ((rc = parse_mixpdf(s, r, &s->cg[cs]))) { free(sw); return rc; }
      }
      /* pruned components (no <MIXTURE> entry): weight 0 with an empty Gaussian (GetStream gives them EmptyMixPDF) */
      for (int m = 0; m < Ms; m++)
         if (s->cg[cs + m] < 0) {
            GROW(s->gconst, s->nG, s->capG, 1, float);
            { const int cap = s->capG;
              s->mean = (float *)realloc(s->mean, sizeof(float) * (size_t)cap * s->vecSize);
              s->var = (float *)realloc(s->var, sizeof(float) * (size_t)cap * s->vecSize);
              s->hasG = (unsigned char *)realloc(s->hasG, (size_t)cap); }
            for (int i = 0; i < s->vecSize; i++) {
               s->mean[(size_t)s->nG * s->vecSize + i] = 0.0f;
               s->var[(size_t)s->nG * s->vecSize + i] = (S > 1 && s->dimStream && s->dimStream[i] != stream) ? INFINITY : 1.0f;
            }
            s->gconst[s->nG] = 0.0f; s->hasG[s->nG] = 0;
            s->cg[cs + m] = s->nG++;
         }
      k = rd_next(r);
      if (!(k == T_KEY && !strcmp(r->tok, "STREAM"))) { rd_push(r); break; }
   }
   s->curStream = 0;
   for (int i = 0; i < S; i++) if (!seen[i]) { free(sw); return fail(r, "a stream of the state is not defined"); }
   s->nComp += M;
   int dur = -1;
   if ((rc = parse_duration_opt(s, r, &dur))) { free(sw); return rc; }      /* GetStateInfo HModel.c:1985 */
   mmf_state *st = &s->st[s->nSt];
   st->name = name; st->nMix = M; st->comp0 = c0; st->inlineOwner = -1; st->src = s->nFiles; st->sMix = NULL; st->sw = sw; st->dur = dur;
   if (S > 1) { st->sMix = (int *)malloc(sizeof(int) * (size_t)S); memcpy(st->sMix, nMixS, sizeof(int) * (size_t)S); }
   *sOut = s->nSt++;
   return HTKAMD_OK;
}

static int find_state(const struct htkamd_mmf *s, const char *name)
{
   for (int i = 0; i < s->nSt; i++) if (s->st[i].name && !strcmp(s->st[i].name, name)) return i;
   return -1;
}
static int find_trans(const struct htkamd_mmf *s, const char *name)
{
   for (int i = 0; i < s->nTr; i++) if (s->tr[i].name && !strcmp(s->tr[i].name, name)) return i;
   return -1;
}
static int find_hmm(const struct htkamd_mmf *s, const char *name)
{
   for (int i = 0; i < s->nHm; i++) if (!strcmp(s->hm[i].name, name)) return i;
   return -1;
}

static int parse_transp_body(struct htkamd_mmf *s, rd *r, char *name, int *tOut)
{
   int N, rc;
   if ((rc = rd_int(r, &N))) return rc;
   if (N < 3) return fail(r, "<TRANSP> needs at least 3 states");
   GROW(s->tr, s->nTr, s->capTr, 1, mmf_trans);
   GROW(s->tp, s->nTp, s->capTp, N * N, float);
   for (int i = 0; i < N * N; i++) {
      float x;
      if ((rc = rd_float(r, &x))) return rc;
      s->tp[s->nTp + i] = (x <= MINLARG) ? (float)LZERO : (float)log((double)x);     /* GetTransMat, HModel.c:1965-1975 */
   }
   mmf_trans *t = &s->tr[s->nTr];
   t->name = name; t->N = N; t->off = s->nTp; t->src = s->nFiles;
   s->nTp += N * N;
   *tOut = s->nTr++;
   return HTKAMD_OK;
}

static int parse_hmm(struct htkamd_mmf *s, rd *r, char *name)
{
   int rc, N, k;
   if (find_hmm(s, name) >= 0) { fail(r, "HMM defined twice"); free(name); return HTKAMD_EMODEL; }
   if (rd_next(r) != T_KEY || strcmp(r->tok, "BEGINHMM")) { free(name); return fail(r, "<BEGINHMM> expected"); }
   if ((rc = parse_options(s, r))) { free(name); return rc; }
   if (rd_next(r) != T_KEY || strcmp(r->tok, "NUMSTATES")) { free(name); return fail(r, "<NUMSTATES> expected"); }
   if ((rc = rd_int(r, &N))) { free(name); return rc; }
   if (N < 3) { free(name); return fail(r, "<NUMSTATES> < 3"); }
   int *states = (int *)malloc(sizeof(int) * (size_t)N);
   for (int i = 0; i < N; i++) states[i] = -1;
   int trans = -1, hdur = -1;
   for (;;) {
      k = rd_next(r);
      if (trans >= 0 && ((k == T_KEY && !strcmp(r->tok, "DURATION")) || (k == T_MACRO && r->tok[0] == 'd'))) {      /* GetHMMDef HModel.c:2136: behind the transition matrix */
         rd_push(r);
         if ((rc = parse_duration_opt(s, r, &hdur))) goto bad;
         continue;
      }
      if (k == T_KEY && !strcmp(r->tok, "STATE")) {
         int i;
         if ((rc = rd_int(r, &i))) goto bad;
         if (i < 2 || i > N - 1) { rc = fail(r, "state index out of range"); goto bad; }
         k = rd_next(r);
         if (k == T_MACRO && r->tok[0] == 'w') {         /* an inline state that opens with its ~w stream weights */
            rd_push(r);
            if ((rc = parse_state_body(s, r, NULL, &states[i - 1]))) goto bad;
            s->st[states[i - 1]].inlineOwner = s->nHm;
         } else if (k == T_MACRO) {
            if (r->tok[0] != 's') { rc = fail(r, "~s expected"); goto bad; }
            char *nm;
            if ((rc = rd_name(r, &nm))) goto bad;
            states[i - 1] = find_state(s, nm);
            if (states[i - 1] < 0) { rc = fail(r, "undefined ~s macro"); free(nm); goto bad; }
            free(nm);
         } else {
            rd_push(r);
            if ((rc = parse_state_body(s, r, NULL, &states[i - 1]))) goto bad;
            s->st[states[i - 1]].inlineOwner = s->nHm;
         }
      } else if (k == T_MACRO && r->tok[0] == 't') {
         char *nm;
         if ((rc = rd_name(r, &nm))) goto bad;
         trans = find_trans(s, nm);
         if (trans < 0) { rc = fail(r, "undefined ~t macro"); free(nm); goto bad; }
         free(nm);
      } else if (k == T_KEY && !strcmp(r->tok, "TRANSP")) {
         if ((rc = parse_transp_body(s, r, NULL, &trans))) goto bad;
      } else if (k == T_KEY && !strcmp(r->tok, "ENDHMM")) break;
      else { rc = fail(r, "unexpected token in HMM definition"); goto bad; }
   }
   if (trans < 0) { rc = fail(r, "HMM without a transition matrix"); goto bad; }
   if (s->tr[trans].N != N) { rc = fail(r, "<TRANSP> size differs from <NUMSTATES>"); goto bad; }
   for (int i = 1; i < N - 1; i++) if (states[i] < 0) { rc = fail(r, "missing <STATE>"); goto bad; }
   GROW(s->hm, s->nHm, s->capHm, 1, mmf_hmm);
   mmf_hmm *h = &s->hm[s->nHm++];
   h->name = name; h->N = N; h->state = states; h->trans = trans; h->src = s->nFiles; h->dur = hdur;
   return HTKAMD_OK;
bad:
   free(states); free(name);
   return rc;
}

static char *base_name(const char *path)
{
   const char *b = strrchr(path, '/');
   char *n = strdup(b ? b + 1 : path);
   char *dot = strrchr(n, '.');
   if (dot && dot != n) *dot = 0;
   return n;
}

static char **g_sortNames;
static int cmp_log_name(const void *a, const void *b)
{
   const int x = *(const int *)a, y = *(const int *)b, c = strcmp(g_sortNames[x], g_sortNames[y]);
   return c ? c : (x > y) - (x < y);
}
int htkamd_mmf_create(struct htkamd_mmf **out)
{
   if (!out) { htkamd_set_error("mmf_create: NULL"); return HTKAMD_EINVAL; }
   *out = (struct htkamd_mmf *)calloc(1, sizeof(struct htkamd_mmf));
   return HTKAMD_OK;
}

/* `defName`: name given to a definition that starts at <BEGINHMM> without a ~h header (a -d directory file); NULL = file base name */
int htkamd_mmf_read(struct htkamd_mmf *s, const char *path, const char *defName)
{
   if (!s || !path) { htkamd_set_error("mmf_read: NULL argument"); return HTKAMD_EINVAL; }
   if (s->finished) { htkamd_set_error("mmf_read: set already finished"); return HTKAMD_EINVAL; }
   rd r; memset(&r, 0, sizeof(r));
   r.f = fopen(path, "rb"); r.path = path; r.line = 1;
   if (!r.f) { htkamd_set_error("mmf_read: cannot open %s", path); return HTKAMD_EIO; }
   setvbuf(r.f, NULL, _IOFBF, 1 << 20);
   int rc = HTKAMD_OK;
   for (;;) {
      int k = rd_next(&r);
      if (k == T_EOF) break;
      if (k == T_KEY && !strcmp(r.tok, "BEGINHMM")) {
         rd_push(&r);
         if ((rc = parse_hmm(s, &r, defName ? strdup(defName) : base_name(path)))) break;
         continue;
      }
      if (k != T_MACRO) { rc = fail(&r, "macro (~x) expected"); break; }
      const char type = r.tok[0];
      if (type == 'o') { if ((rc = parse_options(s, &r))) break; continue; }
      char *name;
      if ((rc = rd_name(&r, &name))) break;
      if (type == 'h') { if ((rc = parse_hmm(s, &r, name))) break; }
      else if (type == 's') {
         int si;
         if (find_state(s, name) >= 0) { rc = fail(&r, "~s macro defined twice"); free(name); break; }
         if ((rc = parse_state_body(s, &r, name, &si))) break;
      } else if (type == 't') {
         int ti;
         if (find_trans(s, name) >= 0) { rc = fail(&r, "~t macro defined twice"); free(name); break; }
         if (rd_next(&r) != T_KEY || strcmp(r.tok, "TRANSP")) { rc = fail(&r, "<TRANSP> expected"); free(name); break; }
         if ((rc = parse_transp_body(s, &r, name, &ti))) break;
      } else if (type == 'm') {
         int g;
         if (find_gauss(s, name) >= 0) { rc = fail(&r, "~m macro defined twice"); free(name); break; }
         if (s->nStreams > 1) s->curStream = -1;           /* stream not known yet: resolve_stream */
         rc = parse_mixpdf(s, &r, &g);
         s->curStream = 0;
         if (rc) { free(name); break; }
         if (s->nStreams > 1) pend_set(s, g, s->lastVecN);
         gname_set(s, g, name);
      } else if (type == 'v' || type == 'u') {
         const char *key = (type == 'v') ? "VARIANCE" : "MEAN";
         if (rd_next(&r) != T_KEY || strcmp(r.tok, key)) { rc = fail(&r, type == 'v' ? "<VARIANCE> expected" : "<MEAN> expected"); free(name); break; }
         if (s->vecSize == 0) { rc = fail(&r, "<VECSIZE> must precede ~u / ~v"); free(name); break; }
         if (find_vmacro(s, type, name) >= 0) { rc = fail(&r, "~u / ~v macro defined twice"); free(name); break; }
         float *v = (float *)malloc(sizeof(float) * (size_t)s->vecSize);
         int vstream = -1;
         if (s->nStreams > 1) {                                        /* only the variance floors, one macro per stream: varFloor1..varFloorS */
            if (!(type == 'v' && !strncmp(name, "varFloor", 8) && name[8] >= '1' && name[8] <= '0' + s->nStreams && !name[9])) {
               rc = fail(&r, "~u / ~v macros in a multi-stream set are not supported (but ~v varFloorN)"); free(v); free(name); break;
            }
            vstream = name[8] - '1';
         }
         if ((rc = parse_vector_ms(s, &r, v, vstream < 0 ? 0 : vstream, 0.0f))) { free(v); free(name); break; }
         if (type == 'v' && !strncmp(name, "varFloor", 8)) {          /* the variance floor macro of HCompV -f: a ~v nobody references */
            if (vstream < 0) {
               free(s->varFloor);
               s->varFloor = (float *)malloc(sizeof(float) * (size_t)s->vecSize);
               memcpy(s->varFloor, v, sizeof(float) * (size_t)s->vecSize);
            } else {
               if (!s->varFloor) s->varFloor = (float *)calloc((size_t)s->vecSize, sizeof(float));
               for (int i = 0; i < s->vecSize; i++) if (s->dimStream[i] == vstream) s->varFloor[i] = v[i];
            }
         }
         GROW(s->vm, s->nVm, s->capVm, 1, __typeof__(*s->vm));
         s->vm[s->nVm].type = type; s->vm[s->nVm].name = name; s->vm[s->nVm].v = v; s->vm[s->nVm].src = s->nFiles; s->vm[s->nVm].stream = vstream;
         s->nVm++;
      } else if (type == 'd') {                              /* ~d "name" <DURATION> n v1 .. vn */
         int di, dup = 0;
         for (int q = 0; q < s->nDm; q++) if (s->dm[q].name && !strcmp(s->dm[q].name, name)) dup = 1;
         if (dup) { rc = fail(&r, "~d macro defined twice"); free(name); break; }
         if (rd_next(&r) != T_KEY || strcmp(r.tok, "DURATION")) { rc = fail(&r, "<DURATION> expected"); free(name); break; }
         if ((rc = parse_duration_body(s, &r, name, &di))) { free(name); break; }
      } else if (type == 'w') {                              /* ~w "name" <SWEIGHTS> S w1 .. wS */
         int n = 0;
         const int S = s->nStreams > 1 ? s->nStreams : 1;
         if (rd_next(&r) != T_KEY || strcmp(r.tok, "SWEIGHTS")) { rc = fail(&r, "<SWEIGHTS> expected"); free(name); break; }
         if ((rc = rd_int(&r, &n))) { free(name); break; }
         if (n != S) { rc = fail(&r, "incorrect number of stream weights"); free(name); break; }
         int dup = 0;
         for (int q = 0; q < s->nWm; q++) if (!strcmp(s->wm[q].name, name)) dup = 1;
         if (dup) { rc = fail(&r, "~w macro defined twice"); free(name); break; }
         GROW(s->wm, s->nWm, s->capWm, 1, __typeof__(*s->wm));
         for (int i = 0; i < S && !rc; i++) rc = rd_float(&r, &s->wm[s->nWm].w[i]);
         if (rc) { free(name); break; }
         s->wm[s->nWm].name = name;
         s->nWm++;
      } else { rc = fail(&r, "unsupported macro type"); free(name); break; }
   }
   fclose(r.f);
   s->nFiles++;
   return rc;
}

/* HMM list (MakeHMMSet): "logical [physical]" per line; physical models not yet defined are read from dir/name[.ext] */
int htkamd_mmf_finish(struct htkamd_mmf *s, const char *hmmList, const char *dir, const char *ext)
{
   if (!s) { htkamd_set_error("mmf_finish: NULL"); return HTKAMD_EINVAL; }
   if (s->finished) return HTKAMD_OK;
   if (hmmList) {
      FILE *f = fopen(hmmList, "r");
      if (!f) { htkamd_set_error("mmf_finish: cannot open HMM list %s", hmmList); return HTKAMD_EIO; }
      char line[1024], a[512], b[512];
      int cap = 0;
      while (fgets(line, sizeof(line), f)) {
         int n = sscanf(line, "%511s %511s", a, b);
         if (n < 1) continue;
         char *lo = a, *ph = (n == 2) ? b : a;
         for (int q = 0; q < 2; q++) {                 /* strip quotes */
            char *t = q ? ph : lo; size_t L = strlen(t);
            if (L >= 2 && (t[0] == '"' || t[0] == '\'') && t[L - 1] == t[0]) { memmove(t, t + 1, L - 2); t[L - 2] = 0; }
         }
         int h = find_hmm(s, ph);
         if (h < 0) {
            char path[1400];
            if (ext && *ext) snprintf(path, sizeof(path), "%s/%s.%s", dir ? dir : ".", ph, ext);
            else snprintf(path, sizeof(path), "%s/%s", dir ? dir : ".", ph);
            int rc = htkamd_mmf_read(s, path, ph);
            if (rc) { fclose(f); return rc; }
            h = find_hmm(s, ph);
            if (h < 0) { fclose(f); htkamd_set_error("mmf_finish: %s does not define %s", path, ph); return HTKAMD_EMODEL; }
         }
         if (s->nLog + 1 > cap) { cap = cap * 2 + 64; s->logName = (char **)realloc(s->logName, sizeof(char *) * (size_t)cap); s->logPhys = (int *)realloc(s->logPhys, sizeof(int) * (size_t)cap); }
         s->logName[s->nLog] = strdup(lo); s->logPhys[s->nLog] = h; s->nLog++;
      }
      fclose(f);
   } else {
      s->logName = (char **)malloc(sizeof(char *) * (size_t)(s->nHm ? s->nHm : 1));
      s->logPhys = (int *)malloc(sizeof(int) * (size_t)(s->nHm ? s->nHm : 1));
      for (int h = 0; h < s->nHm; h++) { s->logName[h] = strdup(s->hm[h].name); s->logPhys[h] = h; }
      s->nLog = s->nHm;
   }
   if (s->nHm == 0 || s->vecSize == 0) { htkamd_set_error("mmf_finish: no model defined"); return HTKAMD_EMODEL; }
   if (s->cov[0] == 0) snprintf(s->cov, sizeof(s->cov), "DIAGC");
   if (s->dur[0] == 0) snprintf(s->dur, sizeof(s->dur), "NULLD");
   if (s->streamWidth == 0) s->streamWidth = s->vecSize;
   const int NS = s->nStreams > 1 ? s->nStreams : 1;
   if (NS > 1 && !s->dimStream) { htkamd_set_error("mmf_finish: a multi-stream set without a single Gaussian"); return HTKAMD_EMODEL; }
   /* flat description: one entry per (state, stream) */
   s->stateCompOff = (int *)malloc(sizeof(int) * ((size_t)s->nSt * NS + 1));
   s->gStr = (int *)calloc((size_t)(s->nG ? s->nG : 1), sizeof(int));
   for (int i = 0; i < s->nSt; i++) {
      int c = s->st[i].comp0;
      for (int k = 0; k < NS; k++) {
         s->stateCompOff[(size_t)i * NS + k] = c;
         const int n = NS > 1 ? s->st[i].sMix[k] : s->st[i].nMix;
         for (int m = 0; m < n; m++) s->gStr[s->cg[c + m]] = k;
         c += n;
      }
   }
   s->stateCompOff[(size_t)s->nSt * NS] = s->nComp;
   if (NS > 1)                                         /* a ~m macro nobody used: the first stream of its width */
      for (int g = 0; g < s->nG && g < s->capPend; g++)
         if (s->gPend[g] > 0) {
            int k, hit = -1;
            for (k = 0; k < NS; k++) if (s->swidth[k] == s->gPend[g]) { hit = k; break; }
            if (hit < 0) { htkamd_set_error("mmf_finish: a ~m macro of width %d fits no stream", s->gPend[g]); return HTKAMD_EMODEL; }
            const int D = s->vecSize, n = s->gPend[g];
            float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
            for (int pass = 0; pass < 2; pass++) {
               float *row = (pass ? s->var : s->mean) + (size_t)g * D;
               memcpy(tmp, row, sizeof(float) * (size_t)n);
               int j = 0;
               for (int i = 0; i < D; i++) row[i] = (s->dimStream[i] == hit) ? tmp[j++] : (pass ? INFINITY : 0.0f);
            }
            free(tmp);
            s->gPend[g] = -(hit + 1); s->gStr[g] = hit;
         }
   s->transN = (int *)malloc(sizeof(int) * (size_t)s->nTr);
   s->transOff = (int *)malloc(sizeof(int) * ((size_t)s->nTr + 1));
   for (int t = 0; t < s->nTr; t++) { s->transN[t] = s->tr[t].N; s->transOff[t] = s->tr[t].off; }
   s->transOff[s->nTr] = s->nTp;
   s->hmmTrans = (int *)malloc(sizeof(int) * (size_t)s->nHm);
   s->hmmStateOff = (int *)malloc(sizeof(int) * ((size_t)s->nHm + 1));
   int tot = 0;
   for (int h = 0; h < s->nHm; h++) tot += s->hm[h].N - 2;
   s->hmmState = (int *)malloc(sizeof(int) * (size_t)(tot ? tot : 1));
   tot = 0;
   for (int h = 0; h < s->nHm; h++) {
      s->hmmTrans[h] = s->hm[h].trans; s->hmmStateOff[h] = tot;
      for (int i = 1; i < s->hm[h].N - 1; i++) s->hmmState[tot++] = s->hm[h].state[i];
   }
   s->hmmStateOff[s->nHm] = tot;
   int anyG = 0, allG = 1;
   for (int g = 0; g < s->nG; g++) { if (s->hasG[g]) anyG = 1; else allG = 0; }
   if (anyG && !allG)                                 /* CheckMix: missing gConst computed at load (HModel.c:206-208) */
      for (int g = 0; g < s->nG; g++) if (!s->hasG[g]) htkamd_host_fix_diag_gconst_ms(s->vecSize, s->var + (size_t)g * s->vecSize, NS > 1 ? s->dimStream : NULL, s->gStr[g], s->gconst + g);
   htkamd_model_desc *d = &s->d;
   d->vecSize = s->vecSize; d->numStates = s->nSt; d->numComp = s->nComp; d->numGauss = s->nG; d->numTrans = s->nTr; d->numPhys = s->nHm;
   d->stateCompOff = s->stateCompOff; d->compWeight = s->wt; d->compGauss = s->cg; d->mean = s->mean; d->var = s->var;
   d->gconst = anyG ? s->gconst : NULL;
   d->transN = s->transN; d->transOff = s->transOff; d->transP = s->tp;
   d->hmmTrans = s->hmmTrans; d->hmmStateOff = s->hmmStateOff; d->hmmState = s->hmmState;
   d->numStreams = NS; d->dimStream = NS > 1 ? s->dimStream : NULL;
   d->hsKind = s->tiedMix ? HTKAMD_HS_TIED : HTKAMD_HS_PLAIN;
   d->streamWeight = NULL;
   if (NS > 1) {
      s->swAll = (float *)malloc(sizeof(float) * (size_t)s->nSt * NS);
      for (int i = 0; i < s->nSt; i++) for (int k = 0; k < NS; k++) s->swAll[(size_t)i * NS + k] = s->st[i].sw ? s->st[i].sw[k] : 1.0f;
      d->streamWeight = s->swAll;
   }
   {  /* name index for htkamd_mmf_find_logical */
      g_sortNames = s->logName;
      s->logSorted = (int *)malloc(sizeof(int) * (size_t)(s->nLog ? s->nLog : 1));
      for (int i = 0; i < s->nLog; i++) s->logSorted[i] = i;
      qsort(s->logSorted, (size_t)s->nLog, sizeof(int), cmp_log_name);
   }
   s->finished = 1;
   return HTKAMD_OK;
}

const htkamd_model_desc *htkamd_mmf_desc(const struct htkamd_mmf *s) { return (s && s->finished) ? &s->d : NULL; }
int htkamd_mmf_num_logical(const struct htkamd_mmf *s) { return s ? s->nLog : 0; }
const char *htkamd_mmf_logical_name(const struct htkamd_mmf *s, int i) { return (s && i >= 0 && i < s->nLog) ? s->logName[i] : NULL; }
int htkamd_mmf_logical_phys(const struct htkamd_mmf *s, int i) { return (s && i >= 0 && i < s->nLog) ? s->logPhys[i] : -1; }
const char *htkamd_mmf_phys_name(const struct htkamd_mmf *s, int h) { return (s && h >= 0 && h < s->nHm) ? s->hm[h].name : NULL; }
const char *htkamd_mmf_parm_kind(const struct htkamd_mmf *s) { return s ? s->kind : NULL; }
const float *htkamd_mmf_var_floor(const struct htkamd_mmf *s) { return s ? s->varFloor : NULL; }

/* Sharing of mean / variance vectors between Gaussians (~u / ~v macros): share[g] = a number that Gaussians with the same vector
 * have in common, -1 for a private vector.  Returns the number of Gaussians that share something (0: nothing to honour). */
int htkamd_mmf_sharing(const struct htkamd_mmf *s, int *meanShare, int *varShare)
{
   int n = 0;
   if (!s) return 0;
   for (int g = 0; g < s->nG; g++) {
      const int mu = (g < s->capMac) ? s->gMeanMac[g] : -1, va = (g < s->capMac) ? s->gVarMac[g] : -1;
      if (meanShare) meanShare[g] = mu;
      if (varShare) varShare[g] = va;
      if (mu >= 0 || va >= 0) n++;
   }
   return n;
}
int htkamd_mmf_find_logical(const struct htkamd_mmf *s, const char *name)
{
   if (!s || !name) return -1;
   if (s->logSorted) {                                /* binary search over the names (first of equal names = first in list order) */
      int lo = 0, hi = s->nLog - 1, hit = -1;
      while (lo <= hi) {
         const int mid = (lo + hi) / 2, c = strcmp(s->logName[s->logSorted[mid]], name);
         if (c == 0) { hit = mid; hi = mid - 1; } else if (c < 0) lo = mid + 1; else hi = mid - 1;
      }
      return hit < 0 ? -1 : s->logPhys[s->logSorted[hit]];
   }
   for (int i = 0; i < s->nLog; i++) if (!strcmp(s->logName[i], name)) return s->logPhys[i];
   return -1;
}

void htkamd_mmf_destroy(struct htkamd_mmf *s)
{
   if (!s) return;
   for (int i = 0; i < s->nSt; i++) { free(s->st[i].name); free(s->st[i].sMix); free(s->st[i].sw); }
   free(s->dimStream); free(s->gStr); free(s->gPend); free(s->swAll);
   for (int i = 0; i < 8; i++) free(s->tmName[i]);
   for (int i = 0; i < s->nTr; i++) free(s->tr[i].name);
   for (int i = 0; i < s->nHm; i++) { free(s->hm[i].name); free(s->hm[i].state); }
   for (int i = 0; i < s->nLog; i++) free(s->logName[i]);
   free(s->logSorted);
   for (int g = 0; g < s->capGN; g++) free(s->gName[g]);
   free(s->gName);
   free(s->st); free(s->wt); free(s->cg); free(s->mean); free(s->var); free(s->gconst); free(s->hasG); free(s->tr); free(s->tp); free(s->hm);
   free(s->varFloor); free(s->logName); free(s->logPhys);
   for (int i = 0; i < s->nVm; i++) { free(s->vm[i].name); free(s->vm[i].v); }
   for (int i = 0; i < s->nWm; i++) free(s->wm[i].name);
   free(s->wm);
   for (int i = 0; i < s->nDm; i++) { free(s->dm[i].name); free(s->dm[i].v); }
   free(s->dm);
   free(s->vm); free(s->gMeanMac); free(s->gVarMac);
   free(s->stateCompOff); free(s->transN); free(s->transOff); free(s->hmmTrans); free(s->hmmStateOff); free(s->hmmState);
   free(s);
}

/* ------------------------------------------------------------------------------------------ writer */
static int g_bin;                                  /* writer mode: text or binary (SaveHMMSet's `binary`), set by htkamd_mmf_write* */
static void put_sym(FILE *f, const char *name, int code) { if (g_bin) { fputc(':', f); fputc(code, f); } else fprintf(f, "<%s>", name); }
static void put_short(FILE *f, int v) { if (g_bin) { fputc((v >> 8) & 255, f); fputc(v & 255, f); } else fprintf(f, " %d", v); }
/* printf("%e") of a float without printf: the float is m * 2^e2 exactly (m < 2^24), so the seven digits are the integer nearest to
   m * 2^e2 * 10^(6-k) (ties to even, as the C library rounds), k = the decimal exponent -- exact in 128-bit integers for every normal
   float.  Writes "d.dddddde+XX" to out (at least 16 bytes), returns the length, or 0 for what is left to fprintf (zero, denormals,
   infinities, NaN). */
static int format_e(float v, char *out)
{
   static const unsigned long long p5[28] = {1ULL, 5ULL, 25ULL, 125ULL, 625ULL, 3125ULL, 15625ULL, 78125ULL, 390625ULL, 1953125ULL, 9765625ULL, 48828125ULL, 244140625ULL,
      1220703125ULL, 6103515625ULL, 30517578125ULL, 152587890625ULL, 762939453125ULL, 3814697265625ULL, 19073486328125ULL, 95367431640625ULL, 476837158203125ULL,
      2384185791015625ULL, 11920928955078125ULL, 59604644775390625ULL, 298023223876953125ULL, 1490116119384765625ULL, 7450580596923828125ULL};
   unsigned int u; memcpy(&u, &v, 4);
   const int neg = (int)(u >> 31), be = (int)((u >> 23) & 255);
   if (be == 0 || be == 255) return 0;
   const unsigned long long m = (u & 0x7FFFFFu) | 0x800000u;
   const int e2 = be - 150;                                         /* v = m * 2^e2 */
   int k = (int)(((be - 127) * 1233) >> 12);                        /* floor(log10) to within one: 1233/4096 ~ log10(2) */
   unsigned long long N = 0;
   for (int tries = 0; tries < 3; tries++) {
      const int p = 6 - k;
      unsigned __int128 num, den = 1;
      int sh;                                                        /* value = num * 2^sh / den */
      if (p >= 0) { if (p > 27) { if (p > 54) return 0; num = (unsigned __int128)m * p5[27] * p5[p - 27]; } else num = (unsigned __int128)m * p5[p]; sh = e2 + p; }
      else { const int q = -p; if (q > 40) return 0; num = m; den = (q > 27) ? (unsigned __int128)p5[27] * p5[q - 27] : p5[q]; sh = e2 - q; }
      unsigned __int128 quo, rem, half;
      if (sh >= 0) {
         if (sh > 100) return 0;
         num <<= sh;
         quo = num / den; rem = num % den; half = den;               /* compare 2 rem with den */
         const unsigned __int128 r2 = rem * 2;
         if (r2 > half || (r2 == half && (quo & 1))) quo++;
      } else {
         const int s = -sh;                                          /* num / (den * 2^s), den == 1 whenever p >= 0 */
         if (den != 1) { if (s > 40) return 0; den <<= s; quo = num / den; rem = num % den; const unsigned __int128 r2 = rem * 2; if (r2 > den || (r2 == den && (quo & 1))) quo++; }
         else if (s >= 128) quo = 0;
         else {
            quo = num >> s; rem = num & ((((unsigned __int128)1) << s) - 1); half = ((unsigned __int128)1) << (s - 1);
            if (rem > half || (rem == half && (quo & 1))) quo++;
         }
      }
      if (quo >= 10000000u) { if (quo == 10000000u && tries == 2) { N = 1000000u; k++; break; } k++; continue; }
      if (quo < 1000000u) { k--; continue; }
      N = (unsigned long long)quo;
      break;
   }
   if (N < 1000000u || N >= 10000000u) return 0;
   int n = 0;
   if (neg) out[n++] = '-';
   char d[8];
   for (int i = 6; i >= 0; i--) { d[i] = (char)('0' + N % 10); N /= 10; }
   out[n++] = d[0]; out[n++] = '.';
   for (int i = 1; i < 7; i++) out[n++] = d[i];
   out[n++] = 'e';
   int ke = k;
   if (ke < 0) { out[n++] = '-'; ke = -ke; } else out[n++] = '+';
   if (ke >= 100) return 0;
   out[n++] = (char)('0' + ke / 10); out[n++] = (char)('0' + ke % 10);
   out[n] = 0;
   return n;
}
static void put_float(FILE *f, float v)
{
   if (g_bin) { unsigned int u; memcpy(&u, &v, 4); putc_unlocked((int)(u >> 24), f); putc_unlocked((int)((u >> 16) & 255), f); putc_unlocked((int)((u >> 8) & 255), f); putc_unlocked((int)(u & 255), f); }
   else {
      char b[24];
      const int n = format_e(v, b + 1);
      if (n) { b[0] = ' '; fwrite_unlocked(b, 1, (size_t)n + 1, f); }
      else fprintf(f, " %e", v);
   }
}
static void put_nl(FILE *f) { if (!g_bin) fputc('\n', f); }
static void put_name(FILE *f, char type, const char *name)
{
   /* ReWriteString(.., DBL_QUOTE): quotes always, backslash before quote/backslash */
   fprintf(f, "~%c \"", type);
   for (const char *p = name; *p; p++) { if (*p == '"' || *p == '\\') fputc('\\', f); fputc(*p, f); }
   fprintf(f, "\"");
   put_nl(f);
}
static void put_options(const struct htkamd_mmf *s, FILE *f)
{
   fprintf(f, "~o\n");
   if (s->setId[0]) { put_sym(f, "HMMSETID", 119); fprintf(f, " %s\n", s->setId); }
   put_sym(f, "STREAMINFO", 5);
   if (s->nStreams > 1) { put_short(f, s->nStreams); for (int i = 0; i < s->nStreams; i++) put_short(f, s->swidth[i]); }
   else { put_short(f, 1); put_short(f, s->streamWidth); }
   put_nl(f);
   put_sym(f, "VECSIZE", 6); put_short(f, s->vecSize);
   put_sym(f, s->dur, !strcmp(s->dur, "POISSOND") ? 8 : !strcmp(s->dur, "GAMMAD") ? 9 : !strcmp(s->dur, "RELD") ? 10 : !strcmp(s->dur, "GEND") ? 11 : 7);      /* NULLD unless the set says otherwise */
   fprintf(f, "<%s><%s>", s->kind[0] ? s->kind : "USER", s->cov);      /* parameter and covariance kinds are text even in binary files */
   put_nl(f);
}
static void put_vec(FILE *f, const char *key, int code, const float *v, int n)
{
   put_sym(f, key, code); put_short(f, n); put_nl(f);
   for (int i = 0; i < n; i++) put_float(f, v[i]);
   put_nl(f);
}
/* the vector of one stream out of an undivided row (stream < 0 or one stream: the row itself) */
static void put_vec_ms(const struct htkamd_mmf *s, FILE *f, const char *key, int code, const float *v, int stream)
{
   if (s->nStreams <= 1 || stream < 0) { put_vec(f, key, code, v, s->vecSize); return; }
   put_sym(f, key, code); put_short(f, s->swidth[stream]); put_nl(f);
   for (int i = 0; i < s->vecSize; i++) if (s->dimStream[i] == stream) put_float(f, v[i]);
   put_nl(f);
}
/* PutMixPDF's body (HModel.c:3029): mean and variance inline or as references to their ~u / ~v macros */
static void put_gauss(const struct htkamd_mmf *s, FILE *f, int g, const float *mean, const float *var, const float *gconst)
{
   const int D = s->vecSize;
   const int mu = (g < s->capMac) ? s->gMeanMac[g] : -1, va = (g < s->capMac) ? s->gVarMac[g] : -1;
   if (mu >= 0) put_name(f, 'u', s->vm[mu].name); else put_vec_ms(s, f, "MEAN", 20, mean + (size_t)g * D, s->nStreams > 1 ? s->gStr[g] : -1);
   if (va >= 0) put_name(f, 'v', s->vm[va].name); else put_vec_ms(s, f, "VARIANCE", 21, var + (size_t)g * D, s->nStreams > 1 ? s->gStr[g] : -1);
   if (gconst) { put_sym(f, "GCONST", 24); put_float(f, gconst[g]); put_nl(f); }
}
static void put_dur(const struct htkamd_mmf *s, FILE *f, int d, int define);
static void put_state(const struct htkamd_mmf *s, FILE *f, int si, const float *mean, const float *var, const float *gconst, const float *wt)
{
   const mmf_state *st = &s->st[si];
   const int NS = s->nStreams > 1 ? s->nStreams : 1;
   int needNM = 0, c0 = st->comp0;
   for (int k = 0; k < NS; k++) if ((NS > 1 ? st->sMix[k] : st->nMix) > 1) needNM = 1;
   if (needNM) { put_sym(f, "NUMMIXES", 3); for (int k = 0; k < NS; k++) put_short(f, NS > 1 ? st->sMix[k] : st->nMix); put_nl(f); }
   if (NS > 1) {                                                       /* GetStateInfo gives a multi-stream state weights of 1 when it has none (:1994) */
      put_sym(f, "SWEIGHTS", 19); put_short(f, NS); put_nl(f);
      for (int k = 0; k < NS; k++) put_float(f, st->sw ? st->sw[k] : 1.0f);
      put_nl(f);
   }
   for (int k = 0; k < NS; k++) {
      const int M = NS > 1 ? st->sMix[k] : st->nMix;
      if (NS > 1) { put_sym(f, "STREAM", 18); put_short(f, k + 1); put_nl(f); }
      if (s->tiedMix) {                                                   /* PutTiedMixtures / PutTiedWeights (HModel.c:2706,2620): runs of up to 256 equal weights */
         put_sym(f, "TMIX", 16);
         fprintf(f, " \"");
         for (const char *p = s->tmName[k]; *p; p++) { if (*p == '"' || *p == '\\') fputc('\\', f); fputc(*p, f); }
         fprintf(f, "\"");
         put_nl(f);
         for (int m = 0; m < M; ) {
            int run = 1;
            while (m + run < M && wt[c0 + m + run] == wt[c0 + m] && run < 256) run++;
            if (g_bin) {
               if (run > 1) { put_float(f, wt[c0 + m] - 2.0f); fputc(run, f); } else put_float(f, wt[c0 + m]);
            } else {
               put_float(f, wt[c0 + m]);
               if (run > 1) fprintf(f, "*%d", run);
            }
            m += run;
         }
         put_nl(f);
         c0 += M;
         continue;
      }
      for (int m = 0; m < M; m++) {
         const int c = c0 + m, g = s->cg[c];
         if (!(wt[c] > (float)MINMIX)) continue;                       /* PutStateInfo :3094 */
         if (M > 1) { put_sym(f, "MIXTURE", 17); put_short(f, m + 1); put_float(f, wt[c]); put_nl(f); }
         if (g < s->capGN && s->gName[g]) { put_name(f, 'm', s->gName[g]); continue; }     /* PutMixPDF: macro reference */
         put_gauss(s, f, g, mean, var, gconst);
      }
      c0 += M;
   }
   put_dur(s, f, st->dur, 0);
}
static void put_dur(const struct htkamd_mmf *s, FILE *f, int d, int define)      /* PutDuration HModel.c:2840: a named vector is a reference but where it is defined */
{
   if (d < 0) return;
   if (s->dm[d].name && !define) { put_name(f, 'd', s->dm[d].name); return; }
   put_sym(f, "DURATION", 25); put_short(f, s->dm[d].n); put_nl(f);
   for (int i = 0; i < s->dm[d].n; i++) put_float(f, s->dm[d].v[i]);
   put_nl(f);
}
static void put_trans(FILE *f, const float *logp, int N)
{
   put_sym(f, "TRANSP", 27); put_short(f, N); put_nl(f);
   for (int i = 0; i < N; i++) {
      float row[64], rSum = 0.0f;
      for (int j = 0; j < N; j++) {
         const float x = logp[i * N + j];
         row[j] = (x < (float)LSMALL) ? 0.0f : (float)exp((double)x);       /* L2F */
         rSum += row[j];
      }
      for (int j = 0; j < N; j++) put_float(f, (i == N - 1) ? 0.0f : row[j] / rSum);
      put_nl(f);
   }
}
static void put_hmm(const struct htkamd_mmf *s, FILE *f, int h, int withHdr, const float *mean, const float *var, const float *gconst,
                    const float *wt, const float *tp)
{
   const mmf_hmm *hm = &s->hm[h];
   if (withHdr) put_name(f, 'h', hm->name);
   put_sym(f, "BEGINHMM", 0); put_nl(f);
   put_sym(f, "NUMSTATES", 4); put_short(f, hm->N); put_nl(f);
   for (int i = 1; i < hm->N - 1; i++) {
      put_sym(f, "STATE", 15); put_short(f, i + 1); put_nl(f);
      const int si = hm->state[i];
      if (s->st[si].name) put_name(f, 's', s->st[si].name);
      else put_state(s, f, si, mean, var, gconst, wt);
   }
   if (s->tr[hm->trans].name) put_name(f, 't', s->tr[hm->trans].name);
   else put_trans(f, tp + s->tr[hm->trans].off, hm->N);
   put_dur(s, f, hm->dur, 0);
   put_sym(f, "ENDHMM", 2); put_nl(f);
}

/* hash-table order of SaveMacros (HModel.c Hash :4331 + NewMacro head insertion): same rule as the .acc scan */
static void macro_order(char **names, int n, int *order)
{
   const char **nm = (const char **)malloc(sizeof(char *) * (size_t)(n ? n : 1));
   for (int i = 0; i < n; i++) nm[i] = names[i];
   htkamd_hmm_scan_order(nm, n, order);
   free(nm);
}

/* Text output with the current parameter values (layout of the desc arrays; transP in log form).
 * oneFile != NULL: everything into that file (SaveInOneFile).  Otherwise dir: models that came from a master file go
 * to dir/<base name of nothing>... -- kept simple: one file per physical HMM named dir/<name> (the -d / -M layout). */
int htkamd_mmf_write_binary(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                            const float *transP, const char *oneFile, const char *dir);
static int mmf_write(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                     const float *transP, const char *oneFile, const char *dir);

int htkamd_mmf_write(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                     const float *transP, const char *oneFile, const char *dir)
{
   g_bin = 0;
   return mmf_write(s, mean, var, gconst, compWeight, transP, oneFile, dir);
}

/* The same in HTK's binary form (SaveHMMSet with binary = TRUE, e.g. HERest -B): ':' + code byte for the keywords,
   big-endian shorts and floats, no line structure. */
int htkamd_mmf_write_binary(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                            const float *transP, const char *oneFile, const char *dir)
{
   g_bin = 1;
   const int rc = mmf_write(s, mean, var, gconst, compWeight, transP, oneFile, dir);
   g_bin = 0;
   return rc;
}

/* The macros that came from file `src` (all of them: src < 0) into one file, in SaveHMMSet's order (HModel.c:4342-4470). */
static int write_macros(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                        const float *transP, const char *oneFile, int src)
{
#define SRC(x) (src < 0 || (x) == src)
   {
      FILE *f = fopen(oneFile, "wb");
      if (f) setvbuf(f, NULL, _IOFBF, 1 << 20);
      if (!f) { htkamd_set_error("mmf_write: cannot create %s", oneFile); return HTKAMD_EIO; }
      put_options(s, f);
      int nN = 0;
      char **names = (char **)malloc(sizeof(char *) * (size_t)(s->nSt + s->nTr + s->nHm + 1));
      int *idx = (int *)malloc(sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + 1));
      int *ord = (int *)malloc(sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + 1));
      /* atomic macros, one pass over the hash table (SaveMacros :4351-4380): ~u, ~v and ~t interleaved in its order; a shared
         vector is written with the current value of the first Gaussian that uses it */
      names = (char **)realloc(names, sizeof(char *) * (size_t)(s->nSt + s->nTr + s->nHm + s->nVm + 1));
      idx = (int *)realloc(idx, sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + s->nVm + 1));
      ord = (int *)realloc(ord, sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + s->nVm + 1));
      names = (char **)realloc(names, sizeof(char *) * (size_t)(s->nSt + s->nTr + s->nHm + s->nVm + s->nDm + 1));
      idx = (int *)realloc(idx, sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + s->nVm + s->nDm + 1));
      ord = (int *)realloc(ord, sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + s->nVm + s->nDm + 1));
      const int DUR0 = 1 << 29;                                /* idx >= DUR0: duration macro idx - DUR0 (~d is an atomic macro too, SaveMacros :4375) */
      for (int i = 0; i < s->nVm; i++) if (SRC(s->vm[i].src)) { names[nN] = s->vm[i].name; idx[nN++] = -1 - i; }
      for (int t = 0; t < s->nTr; t++) if (s->tr[t].name && SRC(s->tr[t].src)) { names[nN] = s->tr[t].name; idx[nN++] = t; }
      for (int q = 0; q < s->nDm; q++) if (s->dm[q].name && SRC(s->dm[q].src)) { names[nN] = s->dm[q].name; idx[nN++] = DUR0 + q; }
      macro_order(names, nN, ord);
      for (int k = 0; k < nN; k++) {
         const int t = idx[ord[k]];
         if (t >= DUR0) { put_name(f, 'd', s->dm[t - DUR0].name); put_dur(s, f, t - DUR0, 1); continue; }
         if (t >= 0) { put_name(f, 't', s->tr[t].name); put_trans(f, transP + s->tr[t].off, s->tr[t].N); continue; }
         const int i = -1 - t;
         const float *v = s->vm[i].v;
         for (int g = 0; g < s->nG && g < s->capMac; g++) {
            if (s->vm[i].type == 'u' && s->gMeanMac[g] == i) { v = mean + (size_t)g * s->vecSize; break; }
            if (s->vm[i].type == 'v' && s->gVarMac[g] == i) { v = var + (size_t)g * s->vecSize; break; }
         }
         put_name(f, s->vm[i].type, s->vm[i].name);
         if (s->vm[i].type == 'u') put_vec_ms(s, f, "MEAN", 20, v, s->vm[i].stream); else put_vec_ms(s, f, "VARIANCE", 21, v, s->vm[i].stream);
      }
      nN = 0;                                                  /* ~m macros come after the atomic ones, before the states */
      names = (char **)realloc(names, sizeof(char *) * (size_t)(s->nSt + s->nTr + s->nHm + s->nG + 1));
      idx = (int *)realloc(idx, sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + s->nG + 1));
      ord = (int *)realloc(ord, sizeof(int) * (size_t)(s->nSt + s->nTr + s->nHm + s->nG + 1));
      for (int g = 0; g < s->nG && g < s->capGN; g++) if (s->gName[g] && SRC(s->gSrc[g])) { names[nN] = s->gName[g]; idx[nN++] = g; }
      macro_order(names, nN, ord);
      for (int k = 0; k < nN; k++) {
         const int g = idx[ord[k]];
         put_name(f, 'm', s->gName[g]);
         put_gauss(s, f, g, mean, var, gconst);
      }
      nN = 0;
      for (int i = 0; i < s->nSt; i++) if (s->st[i].name && SRC(s->st[i].src)) { names[nN] = s->st[i].name; idx[nN++] = i; }
      macro_order(names, nN, ord);
      for (int k = 0; k < nN; k++) { const int i = idx[ord[k]]; put_name(f, 's', s->st[i].name); put_state(s, f, i, mean, var, gconst, compWeight); }
      nN = 0;
      for (int h = 0; h < s->nHm; h++) if (SRC(s->hm[h].src)) { names[nN] = s->hm[h].name; idx[nN++] = h; }
      macro_order(names, nN, ord);
      for (int k = 0; k < nN; k++) put_hmm(s, f, idx[ord[k]], 1, mean, var, gconst, compWeight, transP);
      free(names); free(idx); free(ord);
      if (fclose(f)) { htkamd_set_error("mmf_write: write error on %s", oneFile); return HTKAMD_EIO; }
      return HTKAMD_OK;
   }
#undef SRC
}

static int mmf_write(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                     const float *transP, const char *oneFile, const char *dir)
{
   if (!s || !s->finished || !mean || !var || !compWeight || !transP) { htkamd_set_error("mmf_write: bad argument"); return HTKAMD_EINVAL; }
   if (oneFile) return write_macros(s, mean, var, gconst, compWeight, transP, oneFile, -1);
   if (!dir) { htkamd_set_error("mmf_write: neither file nor directory given"); return HTKAMD_EINVAL; }
   for (int i = 0; i < s->nSt; i++) if (s->st[i].name) { htkamd_set_error("mmf_write: a set with ~s macros must be written to one file"); return HTKAMD_EINVAL; }
   for (int t = 0; t < s->nTr; t++) if (s->tr[t].name) { htkamd_set_error("mmf_write: a set with ~t macros must be written to one file"); return HTKAMD_EINVAL; }
   for (int g = 0; g < s->nG && g < s->capGN; g++) if (s->gName[g]) { htkamd_set_error("mmf_write: a set with ~m macros must be written to one file"); return HTKAMD_EINVAL; }
   if (s->nVm > (s->varFloor ? 1 : 0)) { htkamd_set_error("mmf_write: a set with ~u / ~v macros must be written to one file"); return HTKAMD_EINVAL; }
   for (int q = 0; q < s->nDm; q++) if (s->dm[q].name) { htkamd_set_error("mmf_write: a set with ~d macros must be written to one file"); return HTKAMD_EINVAL; }
   for (int h = 0; h < s->nHm; h++) {
      char path[1400];
      snprintf(path, sizeof(path), "%s/%s", dir, s->hm[h].name);
      FILE *f = fopen(path, "w");
      if (!f) { htkamd_set_error("mmf_write: cannot create %s", path); return HTKAMD_EIO; }
      put_options(s, f);                                             /* SAVEGLOBOPTS = TRUE */
      put_hmm(s, f, h, 1, mean, var, gconst, compWeight, transP);
      if (fclose(f)) { htkamd_set_error("mmf_write: write error on %s", path); return HTKAMD_EIO; }
   }
   return HTKAMD_OK;
}

/* SaveHMMSet for a set loaded from SEVERAL master files (HERest -H macros -H hmmdefs -M dir): every macro goes back to the file it
   was loaded from (HModel.c:4388-4470: SaveMacros per MMF), so that the next iteration finds dir/macros and dir/hmmdefs again.
   masterOut[k] = where the k-th file read goes (k = order of the htkamd_mmf_read calls); models that came from files of their own
   beyond those (a -d directory) go to dir/<name>. */
int htkamd_mmf_write_sources(const struct htkamd_mmf *s, const float *mean, const float *var, const float *gconst, const float *compWeight,
                             const float *transP, const char *const *masterOut, int nMaster, const char *dir, int binary)
{
   if (!s || !s->finished || !mean || !var || !compWeight || !transP || nMaster < 0 || (nMaster > 0 && !masterOut)) { htkamd_set_error("mmf_write_sources: bad argument"); return HTKAMD_EINVAL; }
   int rc = HTKAMD_OK;
   g_bin = binary ? 1 : 0;
   for (int k = 0; k < nMaster && !rc; k++) rc = write_macros(s, mean, var, gconst, compWeight, transP, masterOut[k], k);
   for (int h = 0; h < s->nHm && !rc; h++) {
      if (s->hm[h].src < nMaster) continue;
      char path[1400];
      if (!dir) { htkamd_set_error("mmf_write_sources: model %s came from a file of its own and no directory is given", s->hm[h].name); rc = HTKAMD_EINVAL; break; }
      snprintf(path, sizeof(path), "%s/%s", dir, s->hm[h].name);
      rc = write_macros(s, mean, var, gconst, compWeight, transP, path, s->hm[h].src);
   }
   g_bin = 0;
   return rc;
}

/* HCompV's variance floor macro file (PutVFloor, HCompV.c:359-389): one stream. */
int htkamd_mmf_write_vfloors(const char *path, const float *var, int D, float scale)
{
   if (!path || !var || D <= 0) { htkamd_set_error("mmf_write_vfloors: bad argument"); return HTKAMD_EINVAL; }
   FILE *f = fopen(path, "w");
   if (!f) { htkamd_set_error("mmf_write_vfloors: cannot create %s", path); return HTKAMD_EIO; }
   fprintf(f, "~v varFloor1\n<Variance> %d\n", D);
   for (int i = 0; i < D; i++) { float v = var[i]; v *= scale; fprintf(f, " %e", v); }
   fputc('\n', f);
   fclose(f);
   return HTKAMD_OK;
}

/* ------------------------------------------------------------------------------------------ mixture splitting (HHEd MU)
 * MixUpCommand (HHEd.c:4020-4090) on the loaded set: every selected state goes to `target` components (target > 0) or gains
 * -target components (target < 0).  As in the reference: the gConst mean / deviation over the set guard against splitting
 * outliers (SetGCStats :2080, HeaviestMix :2112: score = weight - number of splits, minus 5000 once when gConst < mean - 4 sd);
 * the heaviest component is replaced by a clone with mean + 0.2 sd and half the weight, a second clone with mean - 0.2 sd goes
 * to the end of the state's list (SplitMix :1318, UpMix :2149); defunct components (weight <= MINMIX) are refilled first
 * (FixDefunctMix :2177).  Afterwards the flat description (htkamd_mmf_desc) has the new sizes. */
static int mix_new_gauss(struct htkamd_mmf *s, int from)
{
   const int D = s->vecSize;
   if (s->nG + 1 > s->capG) {
      s->capG = s->capG * 2 + 64;
      s->gconst = (float *)realloc(s->gconst, sizeof(float) * (size_t)s->capG);
      s->mean = (float *)realloc(s->mean, sizeof(float) * (size_t)s->capG * D);
      s->var = (float *)realloc(s->var, sizeof(float) * (size_t)s->capG * D);
      s->hasG = (unsigned char *)realloc(s->hasG, (size_t)s->capG);
   }
   const int g = s->nG++;
   memcpy(s->mean + (size_t)g * D, s->mean + (size_t)from * D, sizeof(float) * (size_t)D);
   memcpy(s->var + (size_t)g * D, s->var + (size_t)from * D, sizeof(float) * (size_t)D);
   s->gconst[g] = s->gconst[from]; s->hasG[g] = 1;
   return g;
}

typedef struct { float w; int g; int hook; } mix_elem;

static int mix_heaviest(struct htkamd_mmf *s, mix_elem *me, int M, float meanGC, float stdGC)
{
   const float gThresh = meanGC - 4.0 * stdGC;
   int maxm = 0;
   float max = me[0].w - me[0].hook;
   if (me[0].hook < 5000 && s->gconst[me[0].g] < gThresh) { max -= 5000.0; me[0].hook = 5000; }
   for (int m = 1; m < M; m++) {
      float w = me[m].w - me[m].hook;
      if (me[m].hook < 5000 && s->gconst[me[m].g] < gThresh) { w -= 5000.0; me[m].hook = 5000; }
      if (w > max) { max = w; maxm = m; }
   }
   return maxm;
}

/* SplitMix: me[m] -> two clones; the first takes its place, the second is returned in *second */
static void mix_split(struct htkamd_mmf *s, mix_elem *me, int m, mix_elem *second, const int *use, int nUse)
{
   const int D = s->vecSize, g0 = me[m].g;
   const int shared = (g0 < s->capGN && s->gName[g0]) || (g0 < nUse && use[g0] > 1);   /* a pdf somebody else uses is left alone (CloneMixPDF); clones made here are private */
   const int g1 = shared ? mix_new_gauss(s, g0) : g0;
   const int g2 = mix_new_gauss(s, g0);
   const int split = me[m].hook + 1;
   const float pertDepth = 0.2;
   for (int k = 0; k < D; k++) {
      const float x = sqrt(s->var[(size_t)g0 * D + k]) * pertDepth;
      const float base = s->mean[(size_t)g0 * D + k];
      s->mean[(size_t)g1 * D + k] = base + x;
      s->mean[(size_t)g2 * D + k] = base - x;
   }
   const float w = me[m].w / 2.0;
   me[m].w = w; me[m].g = g1; me[m].hook = split;
   second->w = w; second->g = g2; second->hook = split;
}

int htkamd_mmf_mixup(struct htkamd_mmf *s, int target, const unsigned char *stateSel)
{
   if (!s || !s->finished || target == 0) { htkamd_set_error("mmf_mixup: bad argument (set not finished, or target 0)"); return HTKAMD_EINVAL; }
   if (s->nStreams > 1) { htkamd_set_error("mmf_mixup: multi-stream sets are not supported"); return HTKAMD_EMODEL; }
   const int D = s->vecSize;
   for (int g = 0; g < s->nG; g++) { htkamd_host_fix_diag_gconst(D, s->var + (size_t)g * D, s->gconst + g); s->hasG[g] = 1; }   /* FixAllGConsts */
   double sum = 0.0, sumsq = 0.0; int count = 0;
   for (int i = 0; i < s->nSt; i++)
      for (int c = s->st[i].comp0; c < s->st[i].comp0 + s->st[i].nMix; c++) { const float x = s->gconst[s->cg[c]]; sum += x; sumsq += x * x; count++; }
   if (!count) { htkamd_set_error("mmf_mixup: empty model set"); return HTKAMD_EMODEL; }
   const float meanGC = sum / count, stdGC = sqrt(sumsq / count - meanGC * meanGC);
   int *use = (int *)calloc((size_t)s->nG + 1, sizeof(int));
   for (int c = 0; c < s->nComp; c++) use[s->cg[c]]++;
   const int nG0 = s->nG;
   float *nwt = NULL; int *ncg = NULL; int nc = 0, capc = 0;
   for (int i = 0; i < s->nSt; i++) {
      const int M = s->st[i].nMix, c0 = s->st[i].comp0;
      int m = M;
      if (!stateSel || stateSel[i]) m = (target < 0) ? M - target : target;
      if (m < M) m = M;                                         /* MU never removes components */
      mix_elem *me = (mix_elem *)malloc(sizeof(mix_elem) * (size_t)m);
      for (int k = 0; k < M; k++) { me[k].w = s->wt[c0 + k]; me[k].g = s->cg[c0 + k]; me[k].hook = 0; }
      int cnt = M;
      if (m > M || (stateSel ? stateSel[i] : 1)) {
         int defunct = 0;
         for (int k = 0; k < M; k++) if (me[k].w <= MINMIX) defunct++;
         if (m > M - defunct) {                                 /* FixDefunctMix: refill dead components first */
            int n2fix = m - M + defunct;
            if (n2fix > defunct) n2fix = defunct;
            for (int f = 0; f < n2fix; f++) {
               int l = 0;
               while (l < M && me[l].w > MINMIX) l++;
               const int hv = mix_heaviest(s, me, M, meanGC, stdGC);
               mix_elem second;
               mix_split(s, me, hv, &second, use, nG0);
               me[l] = second;
            }
         }
         while (cnt < m) {                                      /* UpMix */
            const int hv = mix_heaviest(s, me, cnt, meanGC, stdGC);
            mix_split(s, me, hv, &me[cnt], use, nG0);
            cnt++;
         }
      }
      if (nc + cnt > capc) { capc = (nc + cnt) * 2 + 64; nwt = (float *)realloc(nwt, sizeof(float) * (size_t)capc); ncg = (int *)realloc(ncg, sizeof(int) * (size_t)capc); }
      for (int k = 0; k < cnt; k++) { nwt[nc + k] = me[k].w; ncg[nc + k] = me[k].g; }
      s->st[i].comp0 = nc; s->st[i].nMix = cnt; nc += cnt;
      free(me);
   }
   free(use);
   free(s->wt); free(s->cg);
   s->wt = nwt; s->cg = ncg; s->nComp = nc; s->capComp = capc;
   if (s->nG > s->capGN) {                                      /* names: the clones are un-named */
      s->gName = (char **)realloc(s->gName, sizeof(char *) * (size_t)s->nG);
      for (int g = s->capGN; g < s->nG; g++) s->gName[g] = NULL;
      s->capGN = s->nG;
   }
   for (int i = 0; i < s->nSt; i++) s->stateCompOff[i] = s->st[i].comp0;
   s->stateCompOff[s->nSt] = s->nComp;
   htkamd_model_desc *d = &s->d;
   d->numComp = s->nComp; d->numGauss = s->nG;
   d->compWeight = s->wt; d->compGauss = s->cg; d->mean = s->mean; d->var = s->var; d->gconst = s->gconst;
   return HTKAMD_OK;
}
