/* net.c -- word network (SLF) + dictionary -> the flat model-level recognition network the decoder walks.
 *
 * Replaces, for context-independent model sets (no cross-word / word-internal context expansion), what HVite -w does
 * before recognition: ReadLattice (HNet.c:631-1230, the SLF subset HBuild/HParse write: header, "I=" node lines with
 * W= and v=, "J=" arc lines with S= E= l=), ReadDict (HDict.c:224-287: WORD ['['OUTSYM']'] [PRONPROB] PHONE...), and
 * ExpandWordNet (HNet.c:3438) in its xc == 0 form: every lattice node becomes, per pronunciation, a chain of model
 * nodes ending in a word-end node (CreateIEModels :2710); !NULL / phone-less words become a bare null word node;
 * every lattice arc links the word-end node(s) of its start to the first node of its end with the arc's LM log
 * probability (ProcessCrossWordLinks :2559); an extra null node precedes all initial lattice nodes and another follows
 * all final ones (AddInitialFinal :2180).  Node kinds: HMM (emits, unless its model is a tee model: a_1N > LSMALL),
 * WORD (word end of a real pronunciation: adds the word penalty and pron prob, starts a path record), NULL (passes tokens).
 * When the dictionary is written in phones that are not all model names, pronunciations are expanded word-internally into
 * context-dependent models (resolve_models below).  Sub-lattices, tags and cross-word context expansion are out of this row's scope.
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../csrc/internal.h"

typedef struct { char *word, *outSym; float prob; int nPhones; int *phone; char **phoneName; } dpron;   /* phone = physical model index (filled by resolve_models) */

struct htkamd_net {
   htkamd_net_desc d;
   int *kind, *model, *linkOff, *linkDest, *wordOf;
   float *pronProb, *linkLike;
   char **wordName; int nWordNames;
   dpron *pron; int nPron;
};

static char *rd_word(char **pp)
{
   /* ReadString-style token: optional quotes, backslash escapes */
   char *p = *pp, buf[512]; int n = 0;
   while (isspace((unsigned char)*p)) p++;
   if (!*p) { *pp = p; return NULL; }
   if (*p == '"' || *p == '\'') {
      const char q = *p++;
      while (*p && *p != q && n < 510) { if (*p == '\\' && p[1]) p++; buf[n++] = *p++; }
      if (*p == q) p++;
   } else {
      while (*p && !isspace((unsigned char)*p) && n < 510) { if (*p == '\\' && p[1]) p++; buf[n++] = *p++; }
   }
   buf[n] = 0; *pp = p;
   return strdup(buf);
}


/* ---- phones -> models.  ExpandWordNet (HNet.c:3438) first asks whether every phone of the WHOLE dictionary is a model name
 * (ClosedDict :1876): then pronunciations are chains of exactly those models.  Otherwise the model set defines contexts
 * (DefineContexts :1892) and, cross-word expansion being off by default (ALLOWXWRDEXP = F), pronunciations are expanded
 * WORD-INTERNALLY: phone j gets the model FindModel (:2052) picks for (left context, phone, right context), where the
 * contexts come from the neighbouring phones inside the word (FindLContext/FindRContext :1850-1873), context-free phones
 * (models that never appear as anybody's context, e.g. sp/sil) count as word boundaries (CFWORDBOUNDARY = T), and
 * context-independent phones (only ever defined bare) keep their own name. */
static void tri_strip(const char *name, char *out, size_t n)
{
   const char *b = strchr(name, '-'); b = b ? b + 1 : name;
   snprintf(out, n, "%s", b);
   char *e = strchr(out, '+'); if (e) *e = 0;
}
typedef struct { char **v; int n, cap; } strset;
static int set_has(const strset *s, const char *x) { for (int i = 0; i < s->n; i++) if (!strcmp(s->v[i], x)) return 1; return 0; }
static void set_add(strset *s, const char *x)
{
   if (set_has(s, x)) return;
   if (s->n + 1 > s->cap) { s->cap = s->cap * 2 + 64; s->v = (char **)realloc(s->v, sizeof(char *) * (size_t)s->cap); }
   s->v[s->n++] = strdup(x);
}
static void set_free(strset *s) { for (int i = 0; i < s->n; i++) free(s->v[i]); free(s->v); }

static int resolve_models(dpron *pr, int nPr, const struct htkamd_mmf *hmms, const char *dictPath)
{
   int closed = 1;
   for (int k = 0; k < nPr && closed; k++)
      for (int q = 0; q < pr[k].nPhones; q++)
         if (htkamd_mmf_find_logical(hmms, pr[k].phoneName[q]) < 0) { closed = 0; break; }
   if (closed) {
      for (int k = 0; k < nPr; k++)
         for (int q = 0; q < pr[k].nPhones; q++) pr[k].phone[q] = htkamd_mmf_find_logical(hmms, pr[k].phoneName[q]);
      return HTKAMD_OK;
   }
   /* DefineContexts */
   strset cxs = {0}, dep = {0};                      /* contexts seen; base names that have a context-dependent model */
   int sLeft = 0, sRight = 0;
   const int nLog = htkamd_mmf_num_logical(hmms);
   char buf[1600], base[512];
   for (int i = 0; i < nLog; i++) {
      const char *nm = htkamd_mmf_logical_name(hmms, i);
      const char *mi = strchr(nm, '-'), *pl = strchr(nm, '+');
      if (mi) { snprintf(buf, sizeof(buf), "%.*s", (int)(mi - nm), nm); set_add(&cxs, buf); sLeft = 1; }
      if (pl) { set_add(&cxs, pl + 1); sRight = 1; }
      if (mi || pl) { tri_strip(nm, base, sizeof(base)); set_add(&dep, base); }
   }
   int rc = HTKAMD_OK;
   for (int k = 0; k < nPr && !rc; k++) {
      dpron *d = &pr[k];
      for (int q = 0; q < d->nPhones && !rc; q++) {
         /* contexts: previous / next phone inside the word; a phone that is nobody's context is a word boundary (0) */
         const char *lc = NULL, *rcx = NULL;
         if (q > 0) { tri_strip(d->phoneName[q - 1], base, sizeof(base)); if (set_has(&cxs, base)) lc = d->phoneName[q - 1]; }
         if (q + 1 < d->nPhones) { tri_strip(d->phoneName[q + 1], base, sizeof(base)); if (set_has(&cxs, base)) rcx = d->phoneName[q + 1]; }
         char lcs[256] = "", rcs[256] = "";
         if (lc) tri_strip(lc, lcs, sizeof(lcs));
         if (rcx) tri_strip(rcx, rcs, sizeof(rcs));
         const char *nm = d->phoneName[q];
         const int ci = !set_has(&dep, nm);                   /* IsHCIContextInd: only the bare model exists */
         if ((!lc && !rcx) || ci) snprintf(buf, sizeof(buf), "%s", nm);
         else if ((!lc || !sLeft) && rcx) snprintf(buf, sizeof(buf), "%s+%s", nm, rcs);
         else if ((!rcx || !sRight) && lc) snprintf(buf, sizeof(buf), "%s-%s", lcs, nm);
         else snprintf(buf, sizeof(buf), "%s-%s+%s", lcs, nm, rcs);
         int h = htkamd_mmf_find_logical(hmms, buf);
         if (h < 0) h = htkamd_mmf_find_logical(hmms, nm);      /* "then try the name itself" */
         if (h < 0) { htkamd_set_error("%s: word %s: no model %s (nor %s) in the model set", dictPath, d->word, buf, nm); rc = HTKAMD_EMODEL; }
         d->phone[q] = h;
      }
   }
   set_free(&cxs); set_free(&dep);
   return rc;
}

static int read_dict(const char *path, const struct htkamd_mmf *hmms, dpron **out, int *nOut)
{
   FILE *f = fopen(path, "r");
   if (!f) { htkamd_set_error("net_build: cannot open dictionary %s", path); return HTKAMD_EIO; }
   dpron *pr = NULL; int n = 0, cap = 0, lineNo = 0;
   char line[4096];
   while (fgets(line, sizeof(line), f)) {
      lineNo++;
      char *p = line;
      char *w = rd_word(&p);
      if (!w) continue;
      if (n + 1 > cap) { cap = cap * 2 + 256; pr = (dpron *)realloc(pr, sizeof(dpron) * (size_t)cap); }
      dpron *d = &pr[n];
      memset(d, 0, sizeof(*d));
      d->word = w; d->prob = 0.0f;
      while (isspace((unsigned char)*p)) p++;
      if (*p == '[') {                                       /* output symbol */
         char *e = strchr(p, ']');
         if (!e) { fclose(f); htkamd_set_error("%s:%d: unterminated [outsym]", path, lineNo); return HTKAMD_EMODEL; }
         *e = 0; d->outSym = strdup(p + 1); p = e + 1;
      } else d->outSym = strdup(w);
      int capP = 0;
      for (;;) {
         char *t = rd_word(&p);
         if (!t) break;
         if (d->nPhones == 0) {                              /* optional pronunciation probability (HDict.c:262-275) */
            char *e; double v = strtod(t, &e);
            if (*e == 0 && (isdigit((unsigned char)t[0]) || t[0] == '.')) {
               if (v <= 0.0 || v > 1.0) { fclose(f); htkamd_set_error("%s:%d: pronunciation probability out of range", path, lineNo); return HTKAMD_EMODEL; }
               /* the reference keeps the probability as a FLOAT before taking the log (ReadDictWord's `float v`, NewPron HDict.c:144-150),
                  and treats anything below MINPRONPROB = 1e-6 as log-zero */
               const float vf = (float)v;
               d->prob = (vf >= 1.0E-6f) ? (float)log((double)vf) : (float)LZERO;
               free(t); continue;
            }
         }
         if (d->nPhones + 1 > capP) { capP = capP * 2 + 8; d->phone = (int *)realloc(d->phone, sizeof(int) * (size_t)capP); d->phoneName = (char **)realloc(d->phoneName, sizeof(char *) * (size_t)capP); }
         d->phoneName[d->nPhones] = t; d->phone[d->nPhones] = -1; d->nPhones++;
      }
      n++;
   }
   fclose(f);
   *out = pr; *nOut = n;
   return resolve_models(pr, n, hmms, path);
}

typedef struct { char *word; int var; } lnode;
typedef struct { int s, e; float l; } larc;

static const char *field(const char *line, const char *key, char *buf, size_t nb)
{
   /* value of "key=" in a line of blank-separated fields */
   const size_t kl = strlen(key);
   for (const char *p = line; *p;) {
      while (isspace((unsigned char)*p)) p++;
      if (!strncmp(p, key, kl) && p[kl] == '=') {
         p += kl + 1;
         size_t n = 0;
         if (*p == '"') { p++; while (*p && *p != '"' && n + 1 < nb) buf[n++] = *p++; }
         else while (*p && !isspace((unsigned char)*p) && n + 1 < nb) buf[n++] = *p++;
         buf[n] = 0;
         return buf;
      }
      while (*p && !isspace((unsigned char)*p)) p++;
   }
   return NULL;
}

static int read_slf(const char *path, lnode **nodes, int *nn, larc **arcs, int *na)
{
   FILE *f = fopen(path, "r");
   if (!f) { htkamd_set_error("net_build: cannot open lattice %s", path); return HTKAMD_EIO; }
   char line[4096], v[512];
   int N = -1, L = -1, gotA = 0;
   lnode *ln = NULL; larc *la = NULL;
   while (fgets(line, sizeof(line), f)) {
      const char *p = line;
      while (isspace((unsigned char)*p)) p++;
      if (*p == '#' || !*p) continue;
      if (!strncmp(p, "SUBLAT", 6)) { fclose(f); free(ln); free(la); htkamd_set_error("%s: sub-lattices are not supported", path); return HTKAMD_EMODEL; }
      if (N < 0) {
         if (field(p, "N", v, sizeof(v)) || field(p, "NODES", v, sizeof(v))) {
            N = atoi(v);
            if (!(field(p, "L", v, sizeof(v)) || field(p, "LINKS", v, sizeof(v)))) { fclose(f); htkamd_set_error("%s: N= without L=", path); return HTKAMD_EMODEL; }
            L = atoi(v);
            if (N <= 0 || L < 0) { fclose(f); htkamd_set_error("%s: bad size line", path); return HTKAMD_EMODEL; }
            ln = (lnode *)calloc((size_t)N, sizeof(lnode)); la = (larc *)calloc((size_t)(L ? L : 1), sizeof(larc));
         }
         continue;                                            /* other header fields (VERSION, lmscale, ...) */
      }
      if (p[0] == 'I' && p[1] == '=') {
         const int i = atoi(field(p, "I", v, sizeof(v)));
         if (i < 0 || i >= N) { fclose(f); htkamd_set_error("%s: node index %d out of range", path, i); return HTKAMD_EMODEL; }
         const char *w = field(p, "W", v, sizeof(v));
         if (!w) w = field(p, "WORD", v, sizeof(v));
         free(ln[i].word);
         ln[i].word = strdup(w ? w : "!NULL");
         const char *pv = field(p, "v", v, sizeof(v));
         ln[i].var = pv ? atoi(pv) : 0;
      } else if (p[0] == 'J' && p[1] == '=') {
         const int j = atoi(field(p, "J", v, sizeof(v)));
         if (j < 0 || j >= L) { fclose(f); htkamd_set_error("%s: arc index %d out of range", path, j); return HTKAMD_EMODEL; }
         const char *s = field(p, "S", v, sizeof(v)); if (!s) s = field(p, "START", v, sizeof(v));
         la[j].s = s ? atoi(s) : -1;
         const char *e = field(p, "E", v, sizeof(v)); if (!e) e = field(p, "END", v, sizeof(v));
         la[j].e = e ? atoi(e) : -1;
         const char *l = field(p, "l", v, sizeof(v)); if (!l) l = field(p, "language", v, sizeof(v));
         la[j].l = l ? strtof(l, NULL) : 0.0f;
         if (la[j].s < 0 || la[j].s >= N || la[j].e < 0 || la[j].e >= N) { fclose(f); htkamd_set_error("%s: arc %d has bad end points", path, j); return HTKAMD_EMODEL; }
         gotA++;
      }
   }
   fclose(f);
   if (N < 0) { htkamd_set_error("%s: no N= L= line", path); return HTKAMD_EMODEL; }
   for (int i = 0; i < N; i++) if (!ln[i].word) ln[i].word = strdup("!NULL");
   if (gotA != L) { htkamd_set_error("%s: %d arcs announced, %d read", path, L, gotA); return HTKAMD_EMODEL; }
   *nodes = ln; *nn = N; *arcs = la; *na = L;
   return HTKAMD_OK;
}

void htkamd_net_destroy(struct htkamd_net *n)
{
   if (!n) return;
   free(n->kind); free(n->model); free(n->linkOff); free(n->linkDest); free(n->wordOf); free(n->pronProb); free(n->linkLike);
   for (int i = 0; i < n->nWordNames; i++) free(n->wordName[i]);
   free(n->wordName);
   for (int i = 0; i < n->nPron; i++) {
      for (int q = 0; q < n->pron[i].nPhones; q++) free(n->pron[i].phoneName[q]);
      free(n->pron[i].phoneName); free(n->pron[i].word); free(n->pron[i].outSym); free(n->pron[i].phone);
   }
   free(n->pron);
   free(n);
}

typedef struct { int from, to; float like; } tlink;

static int expand_lattice(lnode *ln, int NN, larc *la, int NA, dpron *pr, int nPr, const char *slfPath, const char *dictPath, const htkamd_model_desc *md, struct htkamd_net **out);

int htkamd_net_build(const char *slfPath, const char *dictPath, const struct htkamd_mmf *hmms, struct htkamd_net **out)
{
   if (!slfPath || !dictPath || !hmms || !out) { htkamd_set_error("net_build: NULL argument"); return HTKAMD_EINVAL; }
   const htkamd_model_desc *md = htkamd_mmf_desc(hmms);
   if (!md) { htkamd_set_error("net_build: model set not finished"); return HTKAMD_EINVAL; }
   dpron *pr = NULL; int nPr = 0, rc;
   if ((rc = read_dict(dictPath, hmms, &pr, &nPr))) return rc;
   lnode *ln = NULL; larc *la = NULL; int NN = 0, NA = 0;
   if ((rc = read_slf(slfPath, &ln, &NN, &la, &NA))) return rc;
   return expand_lattice(ln, NN, la, NA, pr, nPr, slfPath, dictPath, md, out);
}

/* The alignment network of HVite -a (DoAlignment HVite.c:830): LatticeFromLabels (HNet.c:1516) makes the word-level transcription a
   linear lattice -- one node per label, `boundary` (HVite -b) added at both ends when given, arcs without LM score -- which is then
   expanded like any other lattice (all pronunciations of a word in parallel). */
int htkamd_net_build_words(const char *const *words, int nWords, const char *boundary, const char *dictPath,
                           const struct htkamd_mmf *hmms, struct htkamd_net **out)
{
   if (!words || nWords <= 0 || !dictPath || !hmms || !out) { htkamd_set_error("net_build_words: bad argument"); return HTKAMD_EINVAL; }
   if (!htkamd_mmf_desc(hmms)) { htkamd_set_error("net_build_words: model set not finished"); return HTKAMD_EINVAL; }
   dpron *pr = NULL; int nPr = 0, rc;
   if ((rc = read_dict(dictPath, hmms, &pr, &nPr))) return rc;
   const int NN = nWords + (boundary ? 2 : 0), NA = NN - 1;
   lnode *ln = (lnode *)calloc((size_t)NN, sizeof(lnode));
   larc *la = (larc *)calloc((size_t)(NA ? NA : 1), sizeof(larc));
   for (int i = 0; i < NN; i++) {
      const char *w = (boundary && (i == 0 || i == NN - 1)) ? boundary : words[i - (boundary ? 1 : 0)];
      ln[i].word = strdup(w ? w : "!NULL");
      if (i > 0) { la[i - 1].s = i - 1; la[i - 1].e = i; la[i - 1].l = 0.0f; }
   }
   return expand_lattice(ln, NN, la, NA, pr, nPr, "(transcription)", dictPath, htkamd_mmf_desc(hmms), out);
}

static int expand_lattice(lnode *ln, int NN, larc *la, int NA, dpron *pr, int nPr, const char *slfPath, const char *dictPath, const htkamd_model_desc *md, struct htkamd_net **out)
{
   int rc;

   struct htkamd_net *net = (struct htkamd_net *)calloc(1, sizeof(*net));
   net->pron = pr; net->nPron = nPr;
   /* per lattice node: list of (start node, word node) per pronunciation */
   int capN = 0, nN = 0, capL = 0, nL = 0;
   tlink *tl = NULL;
   int *firstOf = (int *)malloc(sizeof(int) * (size_t)NN), *cntOf = (int *)malloc(sizeof(int) * (size_t)NN);
   int *pStart = NULL, *pEnd = NULL; int nInst = 0, capI = 0;
#define NEWNODE(k, m, pp, wd) do { if (nN + 1 > capN) { capN = capN * 2 + 1024; net->kind = (int *)realloc(net->kind, sizeof(int) * (size_t)capN); \
      net->model = (int *)realloc(net->model, sizeof(int) * (size_t)capN); net->pronProb = (float *)realloc(net->pronProb, sizeof(float) * (size_t)capN); \
      net->wordOf = (int *)realloc(net->wordOf, sizeof(int) * (size_t)capN); } \
      net->kind[nN] = (k); net->model[nN] = (m); net->pronProb[nN] = (pp); net->wordOf[nN] = (wd); nN++; } while (0)
#define NEWLINK(a, b, lk) do { if (nL + 1 > capL) { capL = capL * 2 + 4096; tl = (tlink *)realloc(tl, sizeof(tlink) * (size_t)capL); } \
      tl[nL].from = (a); tl[nL].to = (b); tl[nL].like = (lk); nL++; } while (0)

   NEWNODE(HTKAMD_NODE_NULL, -1, 0.0f, -1);                     /* node 0: net->initial */
   NEWNODE(HTKAMD_NODE_NULL, -1, 0.0f, -1);                     /* node 1: net->final   */
   for (int i = 0; i < NN; i++) {
      firstOf[i] = nInst; cntOf[i] = 0;
      const int isNull = !strcmp(ln[i].word, "!NULL");
      /* pronunciations of the word, in the order of the reference's PronHolder list: InitPronHolders (HNet.c:2293) drops a
         pronunciation that repeats an earlier one (same phones, same probability: remDupPron) and PREPENDS each holder, so the list --
         and with it the order of every link made per holder -- runs from the last pronunciation of the dictionary to the first */
      int sel[256], nSel = 0, found = 0;
      for (int k = 0; k < nPr && !isNull; k++) {
         if (strcmp(pr[k].word, ln[i].word)) continue;
         found++;
         if (ln[i].var > 0 && found != ln[i].var) continue;     /* v= selects one pronunciation */
         int dup = 0;
         for (int z = 0; z < nSel && !dup; z++) {
            const dpron *o = &pr[sel[z]];
            if (o->nPhones != pr[k].nPhones || o->prob != pr[k].prob) continue;
            int q = 0;
            while (q < o->nPhones && !strcmp(o->phoneName[q], pr[k].phoneName[q])) q++;
            dup = (q == o->nPhones);
         }
         if (dup) continue;
         if (nSel >= 256) { htkamd_set_error("net_build: word %s has more than 256 pronunciations", ln[i].word); rc = HTKAMD_EMODEL; goto done; }
         sel[nSel++] = k;
      }
      for (int z = nSel - 1; z >= 0; z--) {
         const int k = sel[z];
         if (nInst + 1 > capI) { capI = capI * 2 + 1024; pStart = (int *)realloc(pStart, sizeof(int) * (size_t)capI); pEnd = (int *)realloc(pEnd, sizeof(int) * (size_t)capI); }
         if (pr[k].nPhones == 0) {                              /* phone-less pronunciation: word node only */
            NEWNODE(HTKAMD_NODE_WORD, k, pr[k].prob, k);
            pStart[nInst] = pEnd[nInst] = nN - 1;
         } else {
            int prev = -1;
            for (int q = 0; q < pr[k].nPhones; q++) {
               NEWNODE(HTKAMD_NODE_HMM, pr[k].phone[q], 0.0f, -1);
               if (q == 0) pStart[nInst] = nN - 1; else NEWLINK(prev, nN - 1, 0.0f);
               prev = nN - 1;
            }
            NEWNODE(HTKAMD_NODE_WORD, k, pr[k].prob, k);
            NEWLINK(prev, nN - 1, 0.0f);
            pEnd[nInst] = nN - 1;
         }
         nInst++; cntOf[i]++;
      }
      if (isNull) {
         if (nInst + 1 > capI) { capI = capI * 2 + 1024; pStart = (int *)realloc(pStart, sizeof(int) * (size_t)capI); pEnd = (int *)realloc(pEnd, sizeof(int) * (size_t)capI); }
         NEWNODE(HTKAMD_NODE_NULL, -1, 0.0f, -1);
         pStart[nInst] = pEnd[nInst] = nN - 1;
         nInst++; cntOf[i] = 1;
      } else if (cntOf[i] == 0) {
         htkamd_set_error("net_build: word %s of %s is not in the dictionary %s", ln[i].word, slfPath, dictPath);
         rc = HTKAMD_EMODEL; goto done;
      }
   }
   {
      char *hasPred = (char *)calloc((size_t)NN, 1), *hasFoll = (char *)calloc((size_t)NN, 1);
      for (int j = 0; j < NA; j++) {
         hasPred[la[j].e] = 1; hasFoll[la[j].s] = 1;
         for (int a = 0; a < cntOf[la[j].s]; a++)
            for (int b = 0; b < cntOf[la[j].e]; b++)
               NEWLINK(pEnd[firstOf[la[j].s] + a], pStart[firstOf[la[j].e] + b], la[j].l);
      }
      int nInit = 0, nFin = 0;
      for (int i = 0; i < NN; i++) {
         if (!hasPred[i]) for (int a = 0; a < cntOf[i]; a++) { NEWLINK(0, pStart[firstOf[i] + a], 0.0f); nInit++; }
         if (!hasFoll[i]) for (int a = 0; a < cntOf[i]; a++) { NEWLINK(pEnd[firstOf[i] + a], 1, 0.0f); nFin++; }
      }
      free(hasPred); free(hasFoll);
      if (!nInit || !nFin) { htkamd_set_error("net_build: %s has no initial or no final node", slfPath); rc = HTKAMD_EMODEL; goto done; }
   }
   /* CSR by source node (stable: creation order within a node) */
   net->linkOff = (int *)calloc((size_t)nN + 1, sizeof(int));
   net->linkDest = (int *)malloc(sizeof(int) * (size_t)(nL ? nL : 1));
   net->linkLike = (float *)malloc(sizeof(float) * (size_t)(nL ? nL : 1));
   for (int k = 0; k < nL; k++) net->linkOff[tl[k].from + 1]++;
   for (int i = 0; i < nN; i++) net->linkOff[i + 1] += net->linkOff[i];
   {
      int *fill = (int *)malloc(sizeof(int) * (size_t)nN);
      memcpy(fill, net->linkOff, sizeof(int) * (size_t)nN);
      for (int k = 0; k < nL; k++) { const int at = fill[tl[k].from]++; net->linkDest[at] = tl[k].to; net->linkLike[at] = tl[k].like; }
      free(fill);
   }
   /* "put all n_tr0 nodes first" (ExpandWordNet HNet.c:3632-3645): links to zero-time nodes (word ends, null nodes, tee models) move to
      the front of a node's link array by the reference's own swaps -- HRec walks them first (ReOrderList HRec.c:1152) */
   if (md) {
      for (int n = 0; n < nN; n++) {
         const int l0 = net->linkOff[n], nl = net->linkOff[n + 1] - l0;
#define IS_TR0(x) (net->kind[x] != HTKAMD_NODE_HMM || md->transP[md->transOff[md->hmmTrans[net->model[x]]] + md->transN[md->hmmTrans[net->model[x]]] - 1] > (float)LSMALL)
         for (int a = 0; a < nl; a++) {
            if (IS_TR0(net->linkDest[l0 + a])) continue;
            int b = a + 1;
            while (b < nl && !IS_TR0(net->linkDest[l0 + b])) b++;
            if (b >= nl) break;
            const int td = net->linkDest[l0 + a]; const float tk = net->linkLike[l0 + a];
            net->linkDest[l0 + a] = net->linkDest[l0 + b]; net->linkLike[l0 + a] = net->linkLike[l0 + b];
            net->linkDest[l0 + b] = td; net->linkLike[l0 + b] = tk;
         }
#undef IS_TR0
      }
   }
   net->wordName = (char **)malloc(sizeof(char *) * (size_t)(nPr ? nPr : 1));
   for (int k = 0; k < nPr; k++) net->wordName[k] = strdup(pr[k].outSym);
   net->nWordNames = nPr;
   net->d.nNodes = nN; net->d.nLinks = nL; net->d.nProns = nPr; net->d.initial = 0; net->d.final = 1;
   net->d.kind = net->kind; net->d.model = net->model; net->d.pronProb = net->pronProb;
   net->d.linkOff = net->linkOff; net->d.linkDest = net->linkDest; net->d.linkLike = net->linkLike;
   rc = HTKAMD_OK;
done:
   for (int i = 0; i < NN; i++) free(ln[i].word);
   free(ln); free(la); free(tl); free(firstOf); free(cntOf); free(pStart); free(pEnd);
   if (rc) { htkamd_net_destroy(net); return rc; }
   *out = net;
   return HTKAMD_OK;
}

const htkamd_net_desc *htkamd_net_get(const struct htkamd_net *n) { return n ? &n->d : NULL; }
/* output symbol of pronunciation k (the `model` field of a WORD node); "" = word produces no label */
const char *htkamd_net_out_sym(const struct htkamd_net *n, int k) { return (n && k >= 0 && k < n->nWordNames) ? n->wordName[k] : NULL; }

/* physical models of pronunciation k (after context expansion), in order; returns their number (may exceed `max`) */
int htkamd_net_pron_models(const struct htkamd_net *n, int k, int *models, int max)
{
   if (!n || k < 0 || k >= n->nPron) return -1;
   for (int q = 0; q < n->pron[k].nPhones && q < max; q++) models[q] = n->pron[k].phone[q];
   return n->pron[k].nPhones;
}

/* the dictionary word a pronunciation belongs to (HVite -m/-f label the first model of a word with the word's NAME) */
const char *htkamd_net_word_name(const struct htkamd_net *n, int k) { return (n && k >= 0 && k < n->nPron) ? n->pron[k].word : NULL; }
