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
 * context-dependent models (resolve_models below).  With ALLOWXWRDEXP (htkamd_net_build_ex) the expansion is CROSS-WORD (xc > 0 in
 * ExpandWordNet: CreateX1Model :2773, CreateXEModels :3141, ProcessCrossWordLinks :2559, SetNullContexts :2514) -- expand_xwrd below.
 * Sub-lattices and tags are out of this row's scope.
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
   /* cross-word expansion: the contexts and what is needed to name the models of a pronunciation between two neighbours */
   int xwrd, flags;
   const struct htkamd_mmf *hmms;             /* must outlive the network when xwrd != 0 */
   char **cxName; int nCx;                    /* context c (1-based) = cxName[c-1] */
   char **depName; int nDep;                  /* base names that have context-dependent models */
   int sLeft, sRight;
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

/* HMMSetCxtInfo (DefineContexts HNet.c:1892): the phones that appear as somebody's left / right context, the base names that have
 * context-dependent models, and whether any model carries a left / a right context at all */
typedef struct { strset cxs, dep; int sLeft, sRight; } hci_t;
static void hci_define(hci_t *h, const struct htkamd_mmf *hmms)
{
   char buf[1600], base[512];
   const int nLog = htkamd_mmf_num_logical(hmms);
   memset(h, 0, sizeof(*h));
   for (int i = 0; i < nLog; i++) {
      const char *nm = htkamd_mmf_logical_name(hmms, i);
      const char *mi = strchr(nm, '-'), *pl = strchr(nm, '+');
      if (mi) { snprintf(buf, sizeof(buf), "%.*s", (int)(mi - nm), nm); set_add(&h->cxs, buf); h->sLeft = 1; }
      if (pl) { set_add(&h->cxs, pl + 1); h->sRight = 1; }
      if (mi || pl) { tri_strip(nm, base, sizeof(base)); set_add(&h->dep, base); }
   }
}
static void hci_free(hci_t *h) { set_free(&h->cxs); set_free(&h->dep); }
/* GetHCIContext (HNet.c:1817) with cross-word contexts on: 1-based context number of a phone, -1 = context free */
static int hci_context(const strset *cxs, const char *phone)
{
   char base[512];
   tri_strip(phone, base, sizeof(base));
   for (int i = 0; i < cxs->n; i++) if (!strcmp(cxs->v[i], base)) return i + 1;
   return -1;
}
/* FindModel (HNet.c:2052): the model for `name` between the contexts lc and rc (1-based numbers, <= 0: none); -1 if there is none */
static int find_model_ctx(const struct htkamd_mmf *hmms, const strset *cxs, const strset *dep, int sLeft, int sRight, int flags,
                          int lc, const char *name, int rc, char *tried, size_t nTried)
{
   char buf[1600];
   const int forceCxt = (flags & HTKAMD_NET_FORCECXTEXP) != 0, forceL = (flags & HTKAMD_NET_FORCELEFTBI) != 0, forceR = (flags & HTKAMD_NET_FORCERIGHTBI) != 0;
   const int ci = !set_has(dep, name);                      /* IsHCIContextInd: only the bare model exists */
   if ((lc <= 0 && rc <= 0) || ci) snprintf(buf, sizeof(buf), "%s", name);
   else if ((lc <= 0 || forceR || !sLeft) && rc > 0 && !forceL) snprintf(buf, sizeof(buf), "%s+%s", name, cxs->v[rc - 1]);
   else if ((rc <= 0 || forceL || !sRight) && lc > 0 && !forceR) snprintf(buf, sizeof(buf), "%s-%s", cxs->v[lc - 1], name);
   else if (!forceL && !forceR) snprintf(buf, sizeof(buf), "%s-%s+%s", cxs->v[lc - 1], name, cxs->v[rc - 1]);
   else snprintf(buf, sizeof(buf), "%s", name);
   int h = htkamd_mmf_find_logical(hmms, buf);
   if (h < 0 && (((lc <= 0 && rc <= 0) || !forceCxt) || (lc <= 0 || !forceL) || (rc <= 0 || !forceR)))
      h = htkamd_mmf_find_logical(hmms, name);              /* "then try the name itself" */
   if (tried) snprintf(tried, nTried, "%s", buf);
   return h;
}

/* *xwrd (may be NULL) comes back 1 when the network has to be expanded with cross-word contexts (ExpandWordNet HNet.c:3458-3478) */
static int resolve_models(dpron *pr, int nPr, const struct htkamd_mmf *hmms, const char *dictPath, int flags, int *xwrd)
{
   int closed = 1;
   const int force = (flags & (HTKAMD_NET_FORCECXTEXP | HTKAMD_NET_FORCELEFTBI | HTKAMD_NET_FORCERIGHTBI)) != 0;
   if (xwrd) *xwrd = 0;
   for (int k = 0; k < nPr && closed; k++)
      for (int q = 0; q < pr[k].nPhones; q++)
         if (htkamd_mmf_find_logical(hmms, pr[k].phoneName[q]) < 0) { closed = 0; break; }
   if (closed && !force) {
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
   int rc = HTKAMD_OK, internal = 1;
   const int allowX = (flags & HTKAMD_NET_ALLOWXWRDEXP) != 0 && cxs.n > 0;
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
         if (h < 0 && allowX) internal = 0;                     /* InternalDict (HNet.c:2142) fails: cross-word models are needed */
         else if (h < 0) { htkamd_set_error("%s: word %s: no model %s (nor %s) in the model set", dictPath, d->word, buf, nm); rc = HTKAMD_EMODEL; }
         d->phone[q] = h;
      }
   }
   if (!rc && allowX && (force || !internal) && xwrd) *xwrd = 1;
   else if (!rc && !internal) { htkamd_set_error("%s: the dictionary needs cross-word models", dictPath); rc = HTKAMD_EMODEL; }
   set_free(&cxs); set_free(&dep);
   return rc;
}

static int read_dict(const char *path, const struct htkamd_mmf *hmms, int flags, int *xwrd, dpron **out, int *nOut)
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
   return resolve_models(pr, n, hmms, path, flags, xwrd);
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
      if (!strncmp(p, "SUBLAT", 6)) { fclose(f); free(ln); free(la); htkamd_set_error("%s: a multi-level lattice (SUBLAT) must be expanded before a network is made of it (the reference says the same: InitPronHolders HNet.c:2358, 'Expand lattice before making network'; HBuild -x expands)", path); return HTKAMD_EMODEL; }
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
   for (int i = 0; i < n->nCx; i++) free(n->cxName[i]);
   for (int i = 0; i < n->nDep; i++) free(n->depName[i]);
   free(n->cxName); free(n->depName);
   free(n);
}

typedef struct { int from, to; float like; } tlink;

typedef struct { const struct htkamd_mmf *hmms; int flags; } xwrd_t;
static int expand_lattice(lnode *ln, int NN, larc *la, int NA, dpron *pr, int nPr, const char *slfPath, const char *dictPath, const htkamd_model_desc *md, const xwrd_t *xw, struct htkamd_net **out);

int htkamd_net_build_ex(const char *slfPath, const char *dictPath, const struct htkamd_mmf *hmms, int flags, struct htkamd_net **out)
{
   if (!slfPath || !dictPath || !hmms || !out) { htkamd_set_error("net_build: NULL argument"); return HTKAMD_EINVAL; }
   const htkamd_model_desc *md = htkamd_mmf_desc(hmms);
   if (!md) { htkamd_set_error("net_build: model set not finished"); return HTKAMD_EINVAL; }
   dpron *pr = NULL; int nPr = 0, rc, xwrd = 0;
   if ((rc = read_dict(dictPath, hmms, flags, &xwrd, &pr, &nPr))) return rc;
   lnode *ln = NULL; larc *la = NULL; int NN = 0, NA = 0;
   if ((rc = read_slf(slfPath, &ln, &NN, &la, &NA))) return rc;
   xwrd_t xw; xw.hmms = hmms; xw.flags = flags;
   return expand_lattice(ln, NN, la, NA, pr, nPr, slfPath, dictPath, md, xwrd ? &xw : NULL, out);
}

int htkamd_net_build(const char *slfPath, const char *dictPath, const struct htkamd_mmf *hmms, struct htkamd_net **out)
{
   return htkamd_net_build_ex(slfPath, dictPath, hmms, 0, out);
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
   if ((rc = read_dict(dictPath, hmms, 0, NULL, &pr, &nPr))) return rc;
   const int NN = nWords + (boundary ? 2 : 0), NA = NN - 1;
   lnode *ln = (lnode *)calloc((size_t)NN, sizeof(lnode));
   larc *la = (larc *)calloc((size_t)(NA ? NA : 1), sizeof(larc));
   for (int i = 0; i < NN; i++) {
      const char *w = (boundary && (i == 0 || i == NN - 1)) ? boundary : words[i - (boundary ? 1 : 0)];
      ln[i].word = strdup(w ? w : "!NULL");
      if (i > 0) { la[i - 1].s = i - 1; la[i - 1].e = i; la[i - 1].l = 0.0f; }
   }
   return expand_lattice(ln, NN, la, NA, pr, nPr, "(transcription)", dictPath, htkamd_mmf_desc(hmms), NULL, out);
}

static int expand_lattice(lnode *ln, int NN, larc *la, int NA, dpron *pr, int nPr, const char *slfPath, const char *dictPath, const htkamd_model_desc *md, const xwrd_t *xw, struct htkamd_net **out)
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
   if (xw) {
      /* ---------------- cross-word context expansion (ExpandWordNet with xc > 0) ----------------
       * Every pronunciation instance gets one ENTRY per left context that can reach it (the last context phone of a preceding word,
       * seen through null words; 0 at the start of the lattice) and one word-end node per right context that can follow it (the first
       * context phone of a following word; 0 at the end): first model = FindModel(lc, phone, .), last model = FindModel(., phone, rc),
       * a one-phone word the full (lc, rc) cross-bar (CreateX1Model); context-free phones at either end of a pronunciation (sp, sil
       * where nobody uses it as a context) keep their own model and pass the context on; null words are copied per (lc, rc) pair
       * (ProcessCrossWordLinks).  A word-end node (fc, rc) links only to successors whose first context is rc, entering them at lc = fc,
       * with the arc's LM score.  Models that map to one physical HMM are not merged here (the reference shares them); the paths and
       * their scores are the same. */
      hci_t hc; hci_define(&hc, xw->hmms);
      const int nc = hc.cxs.n, XC = nc + 1;
      int *iLn = NULL, *iPr = NULL, *iP = NULL, *iQ = NULL, *iFc = NULL, *iIc = NULL; int capX = 0;
      unsigned char *LC = (unsigned char *)calloc((size_t)NN * XC, 1), *RC = (unsigned char *)calloc((size_t)NN * XC, 1);
      unsigned char *through = (unsigned char *)calloc((size_t)NN, 1), *hasPredX = (unsigned char *)calloc((size_t)NN, 1), *hasFollX = (unsigned char *)calloc((size_t)NN, 1);
      int *entryOf = NULL, *wendOf = NULL, *crossOf = NULL, *nullOf = NULL, *thruPron = NULL;
      char tried[1600];
      rc = HTKAMD_OK;
      /* pronunciation instances (same selection and order as the word-internal form below) */
      for (int i = 0; i < NN && !rc; i++) {
         firstOf[i] = nInst; cntOf[i] = 0;
         const int isNull = !strcmp(ln[i].word, "!NULL");
         int sel[256], nSel = 0, found = 0;
         for (int k = 0; k < nPr && !isNull; k++) {
            if (strcmp(pr[k].word, ln[i].word)) continue;
            found++;
            if (ln[i].var > 0 && found != ln[i].var) continue;
            int dup = 0;
            for (int z = 0; z < nSel && !dup; z++) {
               const dpron *o = &pr[sel[z]];
               if (o->nPhones != pr[k].nPhones || o->prob != pr[k].prob) continue;
               int q = 0;
               while (q < o->nPhones && !strcmp(o->phoneName[q], pr[k].phoneName[q])) q++;
               dup = (q == o->nPhones);
            }
            if (dup) continue;
            if (nSel >= 256) { htkamd_set_error("net_build: word %s has more than 256 pronunciations", ln[i].word); rc = HTKAMD_EMODEL; break; }
            sel[nSel++] = k;
         }
         if (!isNull && nSel == 0 && !rc) { htkamd_set_error("net_build: word %s of %s is not in the dictionary %s", ln[i].word, slfPath, dictPath); rc = HTKAMD_EMODEL; }
         if (isNull) { through[i] = 1; continue; }
         for (int z = nSel - 1; z >= 0 && !rc; z--) {
            const int k = sel[z];
            if (nInst + 1 > capX) {
               capX = capX * 2 + 1024;
               iLn = (int *)realloc(iLn, sizeof(int) * (size_t)capX); iPr = (int *)realloc(iPr, sizeof(int) * (size_t)capX); iP = (int *)realloc(iP, sizeof(int) * (size_t)capX);
               iQ = (int *)realloc(iQ, sizeof(int) * (size_t)capX); iFc = (int *)realloc(iFc, sizeof(int) * (size_t)capX); iIc = (int *)realloc(iIc, sizeof(int) * (size_t)capX);
            }
            iLn[nInst] = i; iPr[nInst] = k; iP[nInst] = iQ[nInst] = -1; iFc[nInst] = iIc[nInst] = -1;
            if (pr[k].nPhones == 0) through[i] = 1;            /* a word without phones passes the contexts on like !NULL */
            for (int q = 0; q < pr[k].nPhones; q++) {
               const int c = hci_context(&hc.cxs, pr[k].phoneName[q]);
               if (c < 0) continue;
               if (iP[nInst] < 0) { iP[nInst] = q; iIc[nInst] = c; }
               iQ[nInst] = q; iFc[nInst] = c;
            }
            if (pr[k].nPhones > 0 && iP[nInst] < 0) {           /* NewPronHolder HNet.c:2278 */
               htkamd_set_error("net_build: every word must define some context: no phone of %s is a context of the model set", pr[k].word);
               rc = HTKAMD_EMODEL;
            }
            nInst++; cntOf[i]++;
         }
      }
      /* left / right context sets of the lattice nodes, through null words (SetNullContexts) */
      for (int j = 0; j < NA; j++) { hasPredX[la[j].e] = 1; hasFollX[la[j].s] = 1; }
      for (int i = 0; i < NN; i++) { if (!hasPredX[i]) LC[(size_t)i * XC] = 1; if (!hasFollX[i]) RC[(size_t)i * XC] = 1; }
      for (int changed = 1; changed && !rc;) {
         changed = 0;
         for (int j = 0; j < NA; j++) {
            const int a = la[j].s, b = la[j].e;
            for (int x = firstOf[a]; x < firstOf[a] + cntOf[a]; x++)
               if (iFc[x] >= 0 && !LC[(size_t)b * XC + iFc[x]]) { LC[(size_t)b * XC + iFc[x]] = 1; changed = 1; }
            for (int x = firstOf[b]; x < firstOf[b] + cntOf[b]; x++)
               if (iIc[x] >= 0 && !RC[(size_t)a * XC + iIc[x]]) { RC[(size_t)a * XC + iIc[x]] = 1; changed = 1; }
            if (through[a]) for (int c = 0; c < XC; c++) if (LC[(size_t)a * XC + c] && !LC[(size_t)b * XC + c]) { LC[(size_t)b * XC + c] = 1; changed = 1; }
            if (through[b]) for (int c = 0; c < XC; c++) if (RC[(size_t)b * XC + c] && !RC[(size_t)a * XC + c]) { RC[(size_t)a * XC + c] = 1; changed = 1; }
         }
      }
      /* What ProcessCrossWordLinks (HNet.c:2559-2680) makes of the arcs, node by node, as the reference does it:
       *   - a word's END nodes are typed (lc, rc): lc is the word's last context, but 0 on an arc INTO A NULL NODE WITHOUT FOLLOWERS
       *     (:2611) -- such a word has one more end node, `wendFin`, beside those of its other right contexts (NR: the right contexts
       *     its other arcs ask for); the model of right context 0 links to whichever of the two the LAST arc asked for (zeroSrc);
       *   - a null node without followers is ONE node for every context (InitPronHolders :2436-2449, type n_word);
       *   - a null node without predecessors links into every word from its copy (0, 0), whatever the word's first context (:2640),
       *     and only that copy hangs on net->initial (AddInitialFinal :2208);
       *   - otherwise a null node has the copies (lc, rc) its arcs ask for: rc = first context of a following word, or the rcs it
       *     shares with a following null node. */
      int *outOff = (int *)calloc((size_t)NN + 2, sizeof(int)), *outArc = (int *)malloc(sizeof(int) * ((size_t)NA + 1));
      for (int j = 0; j < NA; j++) outOff[la[j].s + 1]++;
      for (int i = 0; i < NN; i++) outOff[i + 1] += outOff[i];
      { int *fill = (int *)calloc((size_t)NN + 1, sizeof(int)); for (int j = 0; j < NA; j++) outArc[outOff[la[j].s] + fill[la[j].s]++] = j; free(fill); }
      unsigned char *NR = (unsigned char *)calloc((size_t)NN * XC, 1), *NF = (unsigned char *)calloc((size_t)NN, 1), *zeroSrc = (unsigned char *)calloc((size_t)NN, 1);
#define FINAL_NULL(b_) (through[b_] && !hasFollX[b_])
      for (int i = 0; i < NN; i++) {
         if (!hasFollX[i]) { NR[(size_t)i * XC] = 1; zeroSrc[i] = 1; }
         for (int z = outOff[i]; z < outOff[i + 1]; z++) {
            const int b = la[outArc[z]].e;
            if (FINAL_NULL(b)) { NF[i] = 1; zeroSrc[i] = 2; }
            else if (through[b]) for (int r = 0; r < XC; r++) if (RC[(size_t)b * XC + r]) { NR[(size_t)i * XC + r] = 1; if (r == 0) zeroSrc[i] = 1; }
            for (int y = firstOf[b]; y < firstOf[b] + cntOf[b]; y++) if (pr[iPr[y]].nPhones > 0 && iIc[y] >= 0) NR[(size_t)i * XC + iIc[y]] = 1;
         }
      }
      int *wendFin = (int *)malloc(sizeof(int) * ((size_t)nInst + 1));
      entryOf = (int *)malloc(sizeof(int) * ((size_t)nInst * XC + 1)); wendOf = (int *)malloc(sizeof(int) * ((size_t)nInst * XC + 1));
      crossOf = (int *)malloc(sizeof(int) * ((size_t)nInst + 1));          /* one-phone words without leading context-free phones: base of the (lc, rc) cross-bar */
      nullOf = (int *)malloc(sizeof(int) * ((size_t)NN + 1));               /* typed copies of a through node: nullOf[i] + lc*XC + rc in nullTab */
      thruPron = (int *)malloc(sizeof(int) * ((size_t)NN + 1));
      int *nullTab = NULL, nNullTab = 0;
      for (size_t z = 0; z < (size_t)nInst * XC; z++) { entryOf[z] = -1; wendOf[z] = -1; }
      int *crossTab = NULL;                                 /* (entry code -2, a cross-bar row entered directly, is no longer produced) */
#define XMODEL(dst, lc_, k_, q_, rc_) do { (dst) = find_model_ctx(xw->hmms, &hc.cxs, &hc.dep, hc.sLeft, hc.sRight, xw->flags, (lc_), pr[k_].phoneName[q_], (rc_), tried, sizeof(tried)); \
         if ((dst) < 0) { htkamd_set_error("net_build: word %s: cannot find hmm %s (GetHCIModel HNet.c:2167)", pr[k_].word, tried); rc = HTKAMD_EMODEL; } } while (0)
      for (int x = 0; x < nInst && !rc; x++) {
         const int i = iLn[x], k = iPr[x], n = pr[k].nPhones, p = iP[x], q = iQ[x];
         crossOf[x] = -1;
         if (n == 0) continue;
         /* contexts inside the word for phone j: the nearest context phones on either side (FindLContext / FindRContext) */
#define LCI(j_, dflt) ({ int c_ = (dflt); for (int z_ = (j_) - 1; z_ >= 0; z_--) { const int t_ = hci_context(&hc.cxs, pr[k].phoneName[z_]); if (t_ >= 0) { c_ = t_; break; } } c_; })
#define RCI(j_, dflt) ({ int c_ = (dflt); for (int z_ = (j_) + 1; z_ < n; z_++) { const int t_ = hci_context(&hc.cxs, pr[k].phoneName[z_]); if (t_ >= 0) { c_ = t_; break; } } c_; })
         /* ---- the reference's sharing rules, so that token sets (N-best: alternatives are told apart by the word-end NODE they came
          * through) and instance counts (-u) see the same nodes:
          *   word ends   one per right context, ONE for all of them when the last context phone does not depend on its right context
          *               (fci, IsRContextInd HNet.c:2101; ProcessCrossWordLinks :2654);
          *   last models one per distinct physical HMM among the right contexts, each followed by its own copies of the trailing
          *               context-free phones, linked to the word ends of the contexts that chose it (CreateXEModels :3152-3200);
          *   first models one per distinct physical HMM among the left contexts, each preceded by its own copies of the leading
          *               context-free phones (:3219-3271);
          *   one context phone (CreateX1Model :2773): a context-independent phone is a single model for every (lc, rc); otherwise every
          *               left context has its own row -- a collating NULL word node (or its copy of the leading context-free phones),
          *               then one model per distinct physical HMM among the right contexts; without any left-context models in the
          *               set (!sLeft) there is a single row. */
         int fci;
         {
            const int lcInt = (p == q) ? -1 : LCI(q, -1);
            if (lcInt < 0) fci = !set_has(&hc.dep, pr[k].phoneName[q]);
            else {
               int first = -2; fci = 1;
               for (int j = 1; j < nc && fci; j++) {             /* sic: the reference's loop stops short of the last context */
                  const int h = find_model_ctx(xw->hmms, &hc.cxs, &hc.dep, hc.sLeft, hc.sRight, xw->flags, lcInt, pr[k].phoneName[q], j, NULL, 0);
                  if (first == -2 || first < 0) first = h; else if (h != first) fci = 0;
               }
            }
         }
         /* word-end nodes */
         int weSingle = -1;
         for (int r = 0; r < XC; r++) {
            if (!NR[(size_t)i * XC + r]) continue;
            if (fci && weSingle >= 0) { wendOf[(size_t)x * XC + r] = weSingle; continue; }
            NEWNODE(HTKAMD_NODE_WORD, k, pr[k].prob, k);
            wendOf[(size_t)x * XC + r] = nN - 1;
            if (fci) weSingle = nN - 1;
         }
         wendFin[x] = -1;
         if (NF[i]) { NEWNODE(HTKAMD_NODE_WORD, k, pr[k].prob, k); wendFin[x] = nN - 1; }
         /* pInst->rc[r]: the word node the model of right context r links to */
#define RCNODE(r_) (((r_) == 0 && zeroSrc[i] == 2) ? wendFin[x] : wendOf[(size_t)x * XC + (r_)])
         /* a model for the last context phone + copies of the trailing context-free phones; returns the head, *last the node that links to the word end */
#define END_CHAIN(head_, last_, lc_, rc_) do { int h_; XMODEL(h_, (lc_), k, q, (rc_)); if (rc) break; \
            NEWNODE(HTKAMD_NODE_HMM, h_, 0.0f, -1); (head_) = (last_) = nN - 1; \
            for (int z_ = q + 1; z_ < n && !rc; z_++) { int hz_; XMODEL(hz_, 0, k, z_, 0); if (rc) break; NEWNODE(HTKAMD_NODE_HMM, hz_, 0.0f, -1); NEWLINK((last_), nN - 1, 0.0f); (last_) = nN - 1; } } while (0)
         int endHmm[XC], endHead[XC], endLast[XC], nEnd = 0;       /* distinct last models of the current row */
#define LINK_ONCE(from_, to_) do { int dup_ = 0; for (int z_ = linkMark; z_ < nL; z_++) if (tl[z_].from == (from_) && tl[z_].to == (to_)) dup_ = 1; if (!dup_) NEWLINK((from_), (to_), 0.0f); } while (0)
         const int linkMark = nL;
         if (p != q) {
            /* CreateXEModels */
            for (int r = 0; r < XC && !rc; r++) {
               if (!RC[(size_t)i * XC + r]) continue;
               int h; XMODEL(h, LCI(q, 0), k, q, r); if (rc) break;
               int e = -1;
               for (int z = 0; z < nEnd; z++) if (endHmm[z] == h) e = z;
               if (e < 0) { e = nEnd++; endHmm[e] = h; END_CHAIN(endHead[e], endLast[e], LCI(q, 0), r); if (rc) break; }
               if (RCNODE(r) >= 0) LINK_ONCE(endLast[e], RCNODE(r));
               if (fci) break;                                     /* "only need to do this once" */
            }
            if (rc) break;
            int midHead = -1, midTail = -1;
            for (int j = p + 1; j < q && !rc; j++) {
               int h; XMODEL(h, LCI(j, 0), k, j, RCI(j, 0)); if (rc) break;
               NEWNODE(HTKAMD_NODE_HMM, h, 0.0f, -1);
               if (midTail >= 0) NEWLINK(midTail, nN - 1, 0.0f); else midHead = nN - 1;
               midTail = nN - 1;
            }
            if (rc) break;
            if (midTail >= 0) for (int z = 0; z < nEnd; z++) NEWLINK(midTail, endHead[z], 0.0f);
            int stHmm[XC], stHead[XC], nSt = 0;
            for (int l = 0; l < XC && !rc; l++) {
               if (!LC[(size_t)i * XC + l]) continue;
               int h; XMODEL(h, l, k, p, RCI(p, 0)); if (rc) break;
               int e = -1;
               for (int z = 0; z < nSt; z++) if (stHmm[z] == h) e = z;
               if (e < 0) {
                  e = nSt++; stHmm[e] = h;
                  NEWNODE(HTKAMD_NODE_HMM, h, 0.0f, -1);
                  int head = nN - 1;
                  if (midHead >= 0) NEWLINK(head, midHead, 0.0f); else for (int z = 0; z < nEnd; z++) NEWLINK(head, endHead[z], 0.0f);
                  for (int z = p - 1; z >= 0 && !rc; z--) { int hz; XMODEL(hz, 0, k, z, 0); if (rc) break; NEWNODE(HTKAMD_NODE_HMM, hz, 0.0f, -1); NEWLINK(nN - 1, head, 0.0f); head = nN - 1; }
                  stHead[e] = head;
               }
               entryOf[(size_t)x * XC + l] = stHead[e];
            }
         } else if (!set_has(&hc.dep, pr[k].phoneName[p])) {
            /* CreateX1Model, context-independent phone: one model for everybody */
            int head = -1, last = -1;
            END_CHAIN(head, last, 0, 0); if (rc) break;
            for (int r = 0; r < XC; r++) if (RC[(size_t)i * XC + r] && RCNODE(r) >= 0) LINK_ONCE(last, RCNODE(r));
            for (int z = p - 1; z >= 0 && !rc; z--) { int hz; XMODEL(hz, 0, k, z, 0); if (rc) break; NEWNODE(HTKAMD_NODE_HMM, hz, 0.0f, -1); NEWLINK(nN - 1, head, 0.0f); head = nN - 1; }
            for (int l = 0; l < XC; l++) if (LC[(size_t)i * XC + l]) entryOf[(size_t)x * XC + l] = head;
         } else {
            /* CreateX1Model: rows of the (lc, rc) cross-bar */
            int rowHead = -1;
            for (int l = 0; l < XC && !rc; l++) {
               if (!LC[(size_t)i * XC + l]) continue;
               if (!hc.sLeft && rowHead >= 0) { entryOf[(size_t)x * XC + l] = rowHead; continue; }      /* one row for all left contexts */
               /* the collating point: a NULL word node, or the row's copy of the leading context-free phones */
               int head, link;
               if (p == 0) { NEWNODE(HTKAMD_NODE_NULL, -1, 0.0f, -1); head = link = nN - 1; }
               else {
                  int hz; XMODEL(hz, 0, k, 0, 0); if (rc) break;
                  NEWNODE(HTKAMD_NODE_HMM, hz, 0.0f, -1); head = link = nN - 1;
                  for (int z = 1; z < p && !rc; z++) { XMODEL(hz, 0, k, z, 0); if (rc) break; NEWNODE(HTKAMD_NODE_HMM, hz, 0.0f, -1); NEWLINK(link, nN - 1, 0.0f); link = nN - 1; }
                  if (rc) break;
               }
               nEnd = 0;
               for (int r = 0; r < XC && !rc; r++) {
                  if (!RC[(size_t)i * XC + r]) continue;
                  int h; XMODEL(h, hc.sLeft ? l : 0, k, q, r); if (rc) break;
                  int e = -1;
                  for (int z = 0; z < nEnd; z++) if (endHmm[z] == h) e = z;
                  if (e < 0) { e = nEnd++; endHmm[e] = h; END_CHAIN(endHead[e], endLast[e], hc.sLeft ? l : 0, r); if (rc) break; NEWLINK(link, endHead[e], 0.0f); }
                  if (RCNODE(r) >= 0) LINK_ONCE(endLast[e], RCNODE(r));
               }
               if (rc) break;
               entryOf[(size_t)x * XC + l] = head;
               rowHead = head;
            }
         }
#undef END_CHAIN
#undef LINK_ONCE
#undef RCNODE
         if (rc) break;
#undef LCI
#undef RCI
      }
      /* typed copies of the null words and of the words without phones: one per (lc, rc) pair that can pass */
      for (int i = 0; i < NN && !rc; i++) {
         nullOf[i] = -1; thruPron[i] = -1;
         if (!through[i]) continue;
         for (int x = firstOf[i]; x < firstOf[i] + cntOf[i]; x++) if (pr[iPr[x]].nPhones == 0) thruPron[i] = iPr[x];
         nullOf[i] = nNullTab;
         nullTab = (int *)realloc(nullTab, sizeof(int) * (size_t)(nNullTab + XC * XC));
#define NEW_THROUGH() do { if (thruPron[i] >= 0) NEWNODE(HTKAMD_NODE_WORD, thruPron[i], pr[thruPron[i]].prob, thruPron[i]); else NEWNODE(HTKAMD_NODE_NULL, -1, 0.0f, -1); } while (0)
         for (int z = 0; z < XC * XC; z++) nullTab[nNullTab + z] = -1;
         if (!hasFollX[i]) {                                    /* one node for every context */
            NEW_THROUGH();
            for (int l = 0; l < XC; l++) nullTab[nNullTab + l * XC] = nN - 1;
         } else {
            unsigned char *need = (unsigned char *)calloc((size_t)XC * XC, 1);
            for (int z = outOff[i]; z < outOff[i + 1]; z++) {
               const int b = la[outArc[z]].e;
               for (int l = 0; l < XC; l++) {
                  if (!LC[(size_t)i * XC + l]) continue;
                  if (through[b]) for (int r = 0; r < XC; r++) if (RC[(size_t)i * XC + r] && RC[(size_t)b * XC + r] && LC[(size_t)b * XC + l]) need[l * XC + r] = 1;
                  for (int y = firstOf[b]; y < firstOf[b] + cntOf[b]; y++)
                     if (pr[iPr[y]].nPhones > 0 && iIc[y] >= 0) need[l * XC + (hasPredX[i] ? iIc[y] : 0)] = 1;
               }
            }
            for (int l = 0; l < XC; l++)
               for (int r = 0; r < XC; r++)
                  if (need[l * XC + r]) { NEW_THROUGH(); nullTab[nNullTab + l * XC + r] = nN - 1; }
            free(need);
         }
#undef NEW_THROUGH
         nNullTab += XC * XC;
      }
      /* links: out of a word end (instance x, right context r) or a typed null copy (l, r) into everything that follows with first context r */
#define LINK_INTO(from_, b_, l_, r_, like_, anyIc_, intoFinal_) do { \
         for (int y_ = firstOf[b_]; y_ < firstOf[b_] + cntOf[b_]; y_++) { \
            if (pr[iPr[y_]].nPhones == 0 || (!(anyIc_) && iIc[y_] != (r_))) continue; \
            const int e_ = entryOf[(size_t)y_ * XC + (l_)]; \
            if (e_ >= 0) NEWLINK((from_), e_, (like_)); \
            else if (e_ == -2) { for (int r2_ = 0; r2_ < XC; r2_++) { const int c_ = crossTab[crossOf[y_] + (l_) * XC + r2_]; if (c_ >= 0) NEWLINK((from_), c_, (like_)); } } \
         } \
         if (through[b_] && ((intoFinal_) || hasFollX[b_])) { const int t_ = nullTab[nullOf[b_] + (l_) * XC + (r_)]; if (t_ >= 0) NEWLINK((from_), t_, (like_)); } \
      } while (0)
      int nInit = 0, nFin = 0;
      for (int j = 0; j < NA && !rc; j++) {
         const int a = la[j].s, b = la[j].e;
         for (int x = firstOf[a]; x < firstOf[a] + cntOf[a]; x++) {
            if (pr[iPr[x]].nPhones == 0) continue;
            if (FINAL_NULL(b)) { if (wendFin[x] >= 0) NEWLINK(wendFin[x], nullTab[nullOf[b]], la[j].l); }
            for (int r = 0; r < XC; r++) {
               const int w = wendOf[(size_t)x * XC + r];
               if (w < 0) continue;
               LINK_INTO(w, b, iFc[x], r, la[j].l, 0, 0);
            }
         }
         if (through[a] && hasFollX[a]) {
            const int init = !hasPredX[a];
            for (int l = 0; l < XC; l++)
               for (int r = 0; r < XC; r++) {
                  const int t = nullTab[nullOf[a] + l * XC + r];
                  if (t < 0) continue;
                  if (init && r != 0) {                           /* copies (0, r) of an initial null node: into following null nodes only */
                     if (through[b]) { const int t2 = nullTab[nullOf[b] + l * XC + r]; if (t2 >= 0) NEWLINK(t, t2, la[j].l); }
                     continue;
                  }
                  LINK_INTO(t, b, l, r, la[j].l, init, 1);
               }
         }
      }
      for (int i = 0; i < NN && !rc; i++) {
         if (!hasPredX[i]) {                                    /* AddInitialFinal: the initial node enters with no left context */
            for (int x = firstOf[i]; x < firstOf[i] + cntOf[i]; x++) {
               if (pr[iPr[x]].nPhones == 0) continue;
               const int e = entryOf[(size_t)x * XC];
               if (e >= 0) { NEWLINK(0, e, 0.0f); nInit++; }
               else if (e == -2) for (int r = 0; r < XC; r++) { const int c = crossTab[crossOf[x] + r]; if (c >= 0) { NEWLINK(0, c, 0.0f); nInit++; } }
            }
            if (through[i]) { const int t = nullTab[nullOf[i]]; if (t >= 0) { NEWLINK(0, t, 0.0f); nInit++; } }      /* FindWordNode(.., n_word): the copy (0, 0) */
         }
         if (!hasFollX[i]) {
            for (int x = firstOf[i]; x < firstOf[i] + cntOf[i]; x++) { const int w = (pr[iPr[x]].nPhones == 0) ? -1 : wendOf[(size_t)x * XC]; if (w >= 0) { NEWLINK(w, 1, 0.0f); nFin++; } }
            if (through[i]) { const int t = nullTab[nullOf[i]]; if (t >= 0) { NEWLINK(t, 1, 0.0f); nFin++; } }
         }
      }
#undef LINK_INTO
#undef FINAL_NULL
#undef XMODEL
      if (!rc && (!nInit || !nFin)) { htkamd_set_error("net_build: %s has no initial or no final node", slfPath); rc = HTKAMD_EMODEL; }
      if (!rc) {
         net->xwrd = 1; net->flags = xw->flags; net->hmms = xw->hmms; net->sLeft = hc.sLeft; net->sRight = hc.sRight;
         net->cxName = hc.cxs.v; net->nCx = hc.cxs.n; net->depName = hc.dep.v; net->nDep = hc.dep.n;
         memset(&hc, 0, sizeof(hc));
      }
      hci_free(&hc);
      free(iLn); free(iPr); free(iP); free(iQ); free(iFc); free(iIc); free(LC); free(RC); free(through); free(hasPredX); free(hasFollX);
      free(entryOf); free(wendOf); free(crossOf); free(nullOf); free(thruPron); free(nullTab); free(crossTab);
      free(outOff); free(outArc); free(NR); free(NF); free(zeroSrc); free(wendFin);
      if (rc) goto done;
   } else {
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

int htkamd_net_is_xwrd(const struct htkamd_net *n) { return n ? n->xwrd : 0; }

/* models of pronunciation k between two neighbours (cross-word networks: the first / last context phone sees the neighbour's) */
int htkamd_net_seq_models(const struct htkamd_net *n, int k, int prevPron, int nextPron, int *models, int max)
{
   if (!n || k < 0 || k >= n->nPron) return -1;
   if (!n->xwrd) return htkamd_net_pron_models(n, k, models, max);
   const dpron *d = &n->pron[k];
   strset cxs, dep; cxs.v = n->cxName; cxs.n = cxs.cap = n->nCx; dep.v = n->depName; dep.n = dep.cap = n->nDep;
   int lcX = 0, rcX = 0, p = -1, q = -1;
   if (prevPron >= 0 && prevPron < n->nPron) for (int z = 0; z < n->pron[prevPron].nPhones; z++) { const int c = hci_context(&cxs, n->pron[prevPron].phoneName[z]); if (c >= 0) lcX = c; }
   if (nextPron >= 0 && nextPron < n->nPron) for (int z = n->pron[nextPron].nPhones - 1; z >= 0; z--) { const int c = hci_context(&cxs, n->pron[nextPron].phoneName[z]); if (c >= 0) rcX = c; }
   for (int z = 0; z < d->nPhones; z++) if (hci_context(&cxs, d->phoneName[z]) >= 0) { if (p < 0) p = z; q = z; }
   for (int j = 0; j < d->nPhones; j++) {
      int lc = 0, rc = 0;
      if (j >= p && j <= q) {
         lc = lcX; rc = rcX;
         for (int z = j - 1; z >= 0; z--) { const int c = hci_context(&cxs, d->phoneName[z]); if (c >= 0) { lc = c; break; } }
         for (int z = j + 1; z < d->nPhones; z++) { const int c = hci_context(&cxs, d->phoneName[z]); if (c >= 0) { rc = c; break; } }
         if (j > p && j < q && hci_context(&cxs, d->phoneName[j]) < 0) { /* context-free phone inside the word: its own model */ }
      }
      const int h = find_model_ctx(n->hmms, &cxs, &dep, n->sLeft, n->sRight, n->flags, lc, d->phoneName[j], rc, NULL, 0);
      if (h < 0) { htkamd_set_error("net_seq_models: no model for phone %s of %s in this context", d->phoneName[j], d->word); return -1; }
      if (j < max) models[j] = h;
   }
   return d->nPhones;
}

/* number of pronunciation k among the pronunciations of its word, 1-based, in dictionary order (Pron.pnum: the v= field of lattices) */
int htkamd_net_pron_num(const struct htkamd_net *n, int k)
{
   if (!n || k < 0 || k >= n->nPron) return 0;
   int v = 1;
   for (int i = 0; i < k; i++) if (!strcmp(n->pron[i].word, n->pron[k].word)) v++;
   return v;
}

/* the dictionary word a pronunciation belongs to (HVite -m/-f label the first model of a word with the word's NAME) */
const char *htkamd_net_word_name(const struct htkamd_net *n, int k) { return (n && k >= 0 && k < n->nPron) ? n->pron[k].word : NULL; }
