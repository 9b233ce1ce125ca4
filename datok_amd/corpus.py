"""Seeded synthetic corpora for the BASELINE.json configs.

All generators return (text: np.uint8[total], doc_off: np.uint64[n_docs+1]).
Constraints kept by construction: valid UTF-8, no U+0004 inside documents,
at least one token per document, no blank run + token anywhere near the
1024-rune window of matrix.go:365.
"""
import numpy as np

_DE_BASE = """
der die das und in den von zu mit sich des auf für ist im dem nicht ein eine als auch es an werden
aus er hat dass sie nach wird bei einer um am sind noch wie einem über einen so zum war haben nur
oder aber vor zur bis mehr durch man sein wurde sei Jahr Jahre Jahren Zeit Stadt Land Haus Mann
Frau Kind Kinder Menschen Leben Welt Tag Tage Woche Monat Abend Morgen Nacht Arbeit Geld Wasser
Straße Schule Universität Regierung Präsident Minister Gesetz Polizei Gericht Richter Anwalt Arzt
Ärztin Krankenhaus Sprache Wörter Bücher Zeitung Geschichte Wissenschaft Forschung Ergebnis
Möglichkeit Schwierigkeit Größe Höhe Länge Stärke Änderung Öffnung Übung Überraschung Gefühl Glück
Müdigkeit Fähigkeit Tätigkeit Bevölkerung Gesellschaft Wirtschaft Unternehmen Geschäft Verkäufer
Käufer Straßenbahn Flughafen Bahnhof Fußball Maßnahme Großstadt Grüße süß heiß weiß groß größer
schön schöner früh früher später spät gut besser beste schlecht klein kleiner neu neue alter alte
junge lange kurze hohe tiefe schnelle langsame wichtige mögliche nötige öffentliche europäische
deutsche französische übliche natürliche tägliche jährliche gehen geht ging gegangen kommen kommt
kam sehen sieht sah machen macht machte sagen sagt sagte geben gibt gab nehmen nimmt nahm finden
findet fand bleiben bleibt blieb stehen steht stand liegen liegt lag bringen bringt brachte denken
denkt dachte wissen weiß wusste müssen muss musste können kann konnte sollen soll sollte wollen
will wollte dürfen darf mögen möchte hören fühlen führen fahren fährt fuhr laufen läuft lief
schreiben schreibt schrieb lesen liest las sprechen spricht sprach arbeiten arbeitet spielen lernen
zeigen helfen kaufen verkaufen öffnen schließen beginnen gewinnen verlieren erklären erzählen
überlegen übernehmen überzeugen ändern prüfen wählen gewählt zählen erhöhen heute gestern morgen
jetzt dann immer nie oft manchmal hier dort oben unten links rechts sehr ganz etwa fast schon
wieder weiter zusammen allein vielleicht natürlich übrigens außerdem trotzdem deshalb dafür dagegen
darüber darunter davor danach während wegen trotz ohne gegen zwischen hinter neben unter wir ihr
ich du mein dein unser euer dieser diese dieses jener welche alle viele einige wenige beide kein
keine nichts etwas jemand niemand wer was wo wann warum wieso weshalb eins zwei drei vier fünf
sechs sieben acht neun zehn elf zwölf zwanzig dreißig hundert tausend erste zweite dritte letzte
nächste Apfel Bäume Blume Garten Wald Berg Fluss See Meer Himmel Sonne Mond Sterne Wetter Regen
Schnee Wind Sommer Winter Frühling Herbst Januar Februar März April Mai Juni Juli August September
Oktober November Dezember Montag Dienstag Mittwoch Donnerstag Freitag Samstag Sonntag Berlin München
Hamburg Köln Frankfurt Stuttgart Düsseldorf Nürnberg Mannheim Zürich Österreich Deutschland Europa
""".split()

_DE_PREFIX = ["Haupt", "Neben", "Stadt", "Land", "Bundes", "Landes", "Welt", "Wirtschafts", "Sprach",
              "Forschungs", "Arbeits", "Schul", "Gesundheits", "Verkehrs", "Umwelt", "Kultur"]
_DE_HEAD = ["amt", "haus", "rat", "plan", "politik", "minister", "bericht", "gesetz", "zentrum", "gebiet",
            "frage", "lösung", "führung", "prüfung", "änderung", "größe", "behörde", "straße", "büro",
            "verband", "förderung", "gruppe", "leitung", "stelle", "kosten", "quelle", "fläche", "würde",
            "mühle", "brücke"]

_DE_ABBR = ["z.B.", "Dr.", "bzw.", "usw.", "Prof.", "ca.", "Nr.", "Abk.", "etc.", "d.h.", "u.a.", "Str."]
_DE_SPECIAL = ["https://korap.ids-mannheim.de/?q=Baum", "www.wikipedia.org", "korap@ids-mannheim.de",
               "10.0.10.51", "5.9.2018", "50.4%", "readme.txt", "3,50", "1998", "12:30", "C&A", "F.D.P.",
               "Ku'damm", "ids-mannheim.de", "info@example.org", "24.12.2021", "2:1", "§ 12", "100 km/h",
               ":-)", ";)", "<b>", "</b>", "18.30 Uhr"]
_DE_END = [".", ".", ".", ".", ".", "?", "!", "…", "...", "!!!", "???"]

_EN_BASE = """
the of and to a in is that it was for on are as with his they at be this from I have or by one had
not but what all were when we there can an your which their said if do will each about how up out
them then she many some so these would other into has more her two like him see time could no make
than first been its who now people my made over did down only way find use may water long little
very after words called just where most know get through back much before go good new write our
used me man too any day same right look think also around another came come work three word must
because does part even place well such here take why things help put years different away again
off went old number great tell men say small every found still between name should Mr home big give
air line set own under read last never us left end along while might next sound below saw something
thought both few those always looked show large often together asked house don't can't won't I'm
we'll they're it's you've he'd isn't wasn't couldn't shouldn't that's there's let's I'll she'll
world going want school important until form food keep children feet land side without boy once
animals life enough took sometimes four head above kind began almost live page got earth need far
hand high year mother light parts country father let night following picture being study second
eyes soon times story boys since white days ever paper hard near sentence better best across
during today others however sure means knew it's try told young miles sun ways thing whole hear
example heard several change answer room sea against top turned learn point city play toward five
using himself usually money seen didn't car morning I've given trees problem complete o'clock
""".split()
_EN_ABBR = ["Mr.", "Mrs.", "Dr.", "e.g.", "i.e.", "U.S.", "Inc.", "vs.", "etc.", "No.", "St."]
_EN_SPECIAL = ["https://example.org/a?b=c", "user@example.com", "3.14", "1,000", "10:45", "AT&T",
               "www.gutenberg.org", "50%", "$5.99", "#5323", "rock'n'roll", "9/11", ":-)"]


def _de_words():
    words = list(dict.fromkeys(_DE_BASE))
    for i, p in enumerate(_DE_PREFIX):
        for j, h in enumerate(_DE_HEAD):
            if (i * 7 + j * 3) % 5 == 0:
                words.append(p + h)
    return words


def _sentence_pool(rng, words, abbr, special, ends, quotes, n_sent, p_abbr, p_special, p_quote, zipf=False):
    """Builds n_sent sentences (bytes). Returns (flat uint8, starts, lens).
    zipf: word i is drawn with probability ~ 1 / (i + 8) instead of uniformly (a long tail of rare types)."""
    wl = np.array(words, dtype=object)
    out = []
    nw = rng.integers(5, 26, size=n_sent)
    cdf = None
    if zipf:
        w = 1.0 / (np.arange(len(wl)) + 8.0)
        cdf = np.cumsum(w / w.sum())
    for i in range(n_sent):
        k = int(nw[i])
        if cdf is None:
            toks = list(wl[rng.integers(0, len(wl), size=k)])
        else:
            toks = list(wl[np.minimum(np.searchsorted(cdf, rng.random(k)), len(wl) - 1)])
        r = rng.random(4)
        if r[0] < p_abbr * k:
            toks[int(rng.integers(0, k))] = abbr[int(rng.integers(0, len(abbr)))]
        if r[1] < p_special * k:
            toks[int(rng.integers(0, k))] = special[int(rng.integers(0, len(special)))]
        toks[0] = toks[0][:1].upper() + toks[0][1:]
        if k > 8 and r[2] < 0.35:
            toks[k // 2] = toks[k // 2] + ","
        s = " ".join(toks) + ends[int(rng.integers(0, len(ends)))]
        if r[3] < p_quote:
            q = quotes[int(rng.integers(0, len(quotes)))]
            s = q[0] + s + q[1]
        out.append(s.encode("utf-8"))
    lens = np.array([len(b) for b in out], dtype=np.int64)
    starts = np.zeros(n_sent, dtype=np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    flat = np.frombuffer(b"".join(out), dtype=np.uint8)
    return flat, starts, lens


def _gather_segments(flat, seg_start, seg_len):
    """Concatenates flat[seg_start[i] : seg_start[i]+seg_len[i]] for all i (vectorised)."""
    total = int(seg_len.sum())
    out_start = np.zeros(len(seg_len), dtype=np.int64)
    np.cumsum(seg_len[:-1], out=out_start[1:])
    idx = np.arange(total, dtype=np.int64) - np.repeat(out_start - seg_start, seg_len)
    return flat[idx]


def _fixed_docs(rng, pool, n_docs, doc_bytes, sep_choices):
    """n_docs documents of exactly doc_bytes bytes: random sentences joined by
    ' ' (or a blank line now and then), cut on a rune boundary, padded."""
    flat, starts, lens = pool
    mean = float(lens.mean()) + 1.5
    per = int(1.25 * doc_bytes / mean) + 12  # sum of `per` sentence lengths must exceed doc_bytes
    out = np.empty(n_docs * doc_bytes, dtype=np.uint8)
    sep_flat = np.frombuffer(b"".join(sep_choices), dtype=np.uint8)
    sep_lens = np.array([len(s) for s in sep_choices], dtype=np.int64)
    sep_starts = np.zeros(len(sep_choices), dtype=np.int64)
    np.cumsum(sep_lens[:-1], out=sep_starts[1:])
    chunk = max(1, (32 << 20) // (doc_bytes + 1))
    for c0 in range(0, n_docs, chunk):
        nd = min(chunk, n_docs - c0)
        sid = rng.integers(0, len(starts), size=(nd, per))
        sep = rng.integers(0, len(sep_choices), size=(nd, per))
        # interleave sentence, separator, sentence, separator ...
        seg_start = np.empty((nd, 2 * per), dtype=np.int64)
        seg_len = np.empty((nd, 2 * per), dtype=np.int64)
        seg_start[:, 0::2] = starts[sid]
        seg_len[:, 0::2] = lens[sid]
        seg_start[:, 1::2] = sep_starts[sep] + len(flat)
        seg_len[:, 1::2] = sep_lens[sep]
        both = np.concatenate([flat, sep_flat])
        rowlen = seg_len.sum(axis=1)
        assert rowlen.min() >= doc_bytes + 4, "raise `per`"
        cat = _gather_segments(both, seg_start.ravel(), seg_len.ravel())
        row0 = np.zeros(nd, dtype=np.int64)
        np.cumsum(rowlen[:-1], out=row0[1:])
        take = row0[:, None] + np.arange(doc_bytes + 4, dtype=np.int64)[None, :]
        win = cat[take]  # nd x (doc_bytes+4)
        body = win[:, :doc_bytes].copy()
        # cut on a rune boundary: if the byte after the cut is a continuation byte,
        # blank the partial rune at the end of the document
        for back in range(0, 3):
            # the rune starting at doc_bytes-1-back needs more than back+1 bytes?
            lead = body[:, doc_bytes - 1 - back]
            need = np.where(lead >= 0xF0, 4, np.where(lead >= 0xE0, 3, np.where(lead >= 0xC0, 2, 1)))
            cut = (need > back + 1) & (lead >= 0xC0)
            for k in range(back + 1):
                body[cut, doc_bytes - 1 - k] = 0x20
        # never end on a dangling blank-only tail longer than a few bytes is fine;
        # make the last byte a full stop so every document ends a sentence
        body[:, doc_bytes - 1] = np.where(body[:, doc_bytes - 1] == 0x20, 0x2E, body[:, doc_bytes - 1])
        out[c0 * doc_bytes:(c0 + nd) * doc_bytes] = body.ravel()
    doc_off = np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(doc_bytes)
    return out, doc_off


def german_docs(n_docs=4096, doc_bytes=4096, seed=2, n_sent=20000):
    """Config 2 / 4 / 5: German-like sentences (5-25 words, umlauts, abbreviations,
    numbers / URLs / e-mails, typographic quotes), exactly doc_bytes per document."""
    rng = np.random.default_rng(seed)
    pool = _sentence_pool(rng, _de_words(), _DE_ABBR, _DE_SPECIAL, _DE_END,
                          [("„", "“"), ("»", "«"), ("\"", "\""), ("‚", "‘")], n_sent,
                          p_abbr=0.03, p_special=0.02, p_quote=0.06)
    return _fixed_docs(rng, pool, n_docs, doc_bytes, [b" ", b" ", b" ", b" ", b" ", b" ", b" ", b"\n\n", b"\n"])


_SYL_ON = ["b", "d", "f", "g", "h", "k", "l", "m", "n", "p", "r", "s", "t", "w", "z", "sch", "st", "sp", "br", "dr", "fr",
           "gr", "kr", "pr", "tr", "bl", "fl", "gl", "kl", "pf", "schw", "str", "v", "j", "qu", "ch", ""]
_SYL_NU = ["a", "e", "i", "o", "u", "ä", "ö", "ü", "ei", "au", "ie", "eu", "äu", "aa", "ee", "oo"]
_SYL_CO = ["", "", "", "n", "r", "l", "s", "t", "m", "ch", "ck", "ng", "nd", "nt", "rt", "st", "ß", "ss", "tz", "ll",
           "mm", "nn", "rr", "ff", "pf", "rz", "lt", "ls", "ns", "cht"]
_SUFFIX = ["", "", "", "en", "er", "e", "ung", "heit", "keit", "lich", "isch", "bar", "los", "schaft", "chen", "lein",
           "ig", "sam", "tum", "nis", "s", "es", "em", "te", "ten", "st", "t"]
_RICH_SPECIAL = ["<p>", "</p>", "<br/>", "<b>", "</b>", "<i>", "</i>", "<a href=\"https://www.example.org/pfad/zur/seite.html?x=1&y=2\">",
                 "</a>", "<span class=\"hervorgehoben\">", "</span>", "<img src=\"bild.png\" alt=\"Ein Bild\"/>", "<h2>", "</h2>",
                 "<li>", "</li>", "<!-- Kommentar -->", "<div id=\"inhalt\">", "</div>",
                 "https://de.wikipedia.org/wiki/Deutsche_Sprache", "http://www.ids-mannheim.de/korap/?q=Baum&ql=poliqarp",
                 "www.example.com/a/b/c/index.php?id=123&lang=de#abschnitt", "max.mustermann@beispiel-firma.de",
                 "info@uni-mannheim.de", "ftp://ftp.example.org/pub/datei.tar.gz", "192.168.178.1", "2001:db8::1",
                 "#hashtag", "@benutzer", "C++", "E-Mail-Adresse", "3,14159", "1.000.000", "12.03.2024", "14:35:07", "§ 823 Abs. 1 BGB",
                 "50 %", "20 °C", "10 km/h", "1/2", "z. B.", "u. a.", "d. h.", "i. d. R.", "Dr. med.", "Prof. Dr.", "Nr. 5", "S. 12 ff.",
                 ":-)", ";-)", ":D", "^^", "…", "–", "—", "(sic!)", "[1]", "{x}", "a/b", "x*y", "1+1=2", "100%ig", "'s", "O'Neill",
                 "Müller-Lüdenscheidt", "Sankt-Nimmerleins-Tag", "AT&T", "H&M", "km²", "CO₂", "µm", "€ 9,99", "$ 5", "£10"]


def _rich_words(n_types, seed):
    rng = np.random.default_rng(seed)
    words, seen = list(_de_words()), set(_de_words())
    while len(words) < n_types:
        k = (1, 2, 2, 2, 3, 3, 4)[int(rng.integers(0, 7))]
        w = "".join(_SYL_ON[int(rng.integers(0, len(_SYL_ON)))] + _SYL_NU[int(rng.integers(0, len(_SYL_NU)))] +
                    _SYL_CO[int(rng.integers(0, len(_SYL_CO)))] for _ in range(k)) + _SUFFIX[int(rng.integers(0, len(_SUFFIX)))]
        r = rng.random()
        if r < 0.35:
            w = w[:1].upper() + w[1:]           # nouns
        elif r < 0.40 and len(words) > 700:
            w = w + words[int(rng.integers(0, 600))]   # compounds with a known head
        elif r < 0.43:
            w = w.upper()                       # acronyms
        elif r < 0.46:
            w = w + "-" + w[::-1].capitalize()  # hyphenated
        if w not in seen:
            seen.add(w); words.append(w)
    return words


def german_rich_docs(n_docs=4096, doc_bytes=4096, seed=2, n_types=30000, n_sent=60000, p_special=0.012):
    """A harder German-like corpus for robustness measurements: `n_types` word types (the built-in list plus
    generated stems, compounds, acronyms, hyphenations), XML tags with attributes, URLs, e-mail addresses,
    numbers with units, abbreviations with blanks inside, emoticons -- one such item per 80 words or so -- and
    typographic quotes.  Same shape as german_docs (exactly doc_bytes per document, valid UTF-8, no U+0004)."""
    rng = np.random.default_rng(seed)
    pool = _sentence_pool(rng, _rich_words(n_types, 7), _DE_ABBR, _DE_SPECIAL + _RICH_SPECIAL, _DE_END,
                          [("„", "“"), ("»", "«"), ("\"", "\""), ("‚", "‘"), ("(", ")"), ("›", "‹")], n_sent,
                          p_abbr=0.02, p_special=p_special, p_quote=0.08, zipf=True)
    return _fixed_docs(rng, pool, n_docs, doc_bytes, [b" ", b" ", b" ", b" ", b" ", b" ", b" ", b"\n\n", b"\n", b"\t", b"  "])


def _german_part(args):
    n_docs, doc_bytes, seed = args
    return german_docs(n_docs, doc_bytes, seed)[0]


def german_docs_sharded(n_docs, doc_bytes=4096, seed=5, parts=16, workers=None):
    """Config 5 (one shard of the 10 GiB corpus is 327 680 x 4 KiB): the same generator run as `parts` independent
    sub-streams (seeds seed * 1000 + part) in a process pool -- the single-threaded generator makes about 10 MB/s.
    The bytes depend on (n_docs, doc_bytes, seed, parts) only, not on the number of workers."""
    import multiprocessing as mp
    import os
    per = [n_docs // parts + (1 if i < n_docs % parts else 0) for i in range(parts)]
    jobs = [(per[i], doc_bytes, seed * 1000 + i) for i in range(parts) if per[i]]
    workers = workers or min(len(jobs), os.cpu_count() or 1)
    if workers <= 1:
        chunks = [_german_part(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:   # spawn: the parent may hold a GPU context
            chunks = pool.map(_german_part, jobs)
    text = np.concatenate(chunks)
    doc_off = np.arange(n_docs + 1, dtype=np.uint64) * np.uint64(doc_bytes)
    return text, doc_off


def english_zipf_docs(n_docs=65536, seed=3, min_bytes=64, max_bytes=65536, n_sent=20000):
    """Config 3: lengths 64*2^k clipped to [64 B, 64 KiB], P(k) ~ 1/(k+1)."""
    rng = np.random.default_rng(seed)
    pool = _sentence_pool(rng, _EN_BASE, _EN_ABBR, _EN_SPECIAL, [".", ".", ".", "?", "!", "..."],
                          [("\"", "\""), ("“", "”"), ("'", "'")], n_sent,
                          p_abbr=0.02, p_special=0.015, p_quote=0.05)
    kmax = int(np.log2(max_bytes // min_bytes))
    pk = 1.0 / (np.arange(kmax + 1) + 1.0)
    pk /= pk.sum()
    ks = rng.choice(kmax + 1, size=n_docs, p=pk)
    parts, lens = [], []
    for k in range(kmax + 1):
        cnt = int((ks == k).sum())
        if cnt == 0:
            continue
        t, _ = _fixed_docs(rng, pool, cnt, min_bytes << k, [b" ", b" ", b" ", b" ", b" ", b"\n\n"])
        parts.append((k, t))
    # scatter back into the drawn order
    doc_len = (min_bytes << ks).astype(np.uint64)
    doc_off = np.zeros(n_docs + 1, dtype=np.uint64)
    np.cumsum(doc_len, out=doc_off[1:])
    text = np.empty(int(doc_off[-1]), dtype=np.uint8)
    for k, t in parts:
        ids = np.flatnonzero(ks == k)
        L = min_bytes << k
        dst = (doc_off[ids].astype(np.int64)[:, None] + np.arange(L, dtype=np.int64)[None, :]).ravel()
        text[dst] = t
    return text, doc_off


def simple_ascii(seed=1, n=1024):
    """Config 1: 1 KiB over letters + the sigma of simpletok (tab, newline, space, ! . ?)."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
    seps = [b" ", b" ", b" ", b"  ", b"\t", b"\n", b"! ", b". ", b"? ", b" -- "]
    out = bytearray()
    while len(out) < n:
        k = int(rng.integers(1, 13))
        out += bytes(letters[rng.integers(0, len(letters), size=k)])
        out += seps[int(rng.integers(0, len(seps)))]
    out = out[:n]
    out[-1:] = b"."
    text = np.frombuffer(bytes(out), dtype=np.uint8).copy()
    return text, np.array([0, n], dtype=np.uint64)


def concat_docs(docs):
    """list[bytes] -> (text, doc_off)"""
    lens = np.array([len(d) for d in docs], dtype=np.uint64)
    doc_off = np.zeros(len(docs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=doc_off[1:])
    text = np.frombuffer(b"".join(docs), dtype=np.uint8).copy() if docs else np.zeros(0, np.uint8)
    return text, doc_off
