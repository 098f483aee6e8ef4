/* hs_tables.h -- embedding constants of the BLOSUM62-metric amino-acid embedding (DATA, not code).
 *
 * The 20x8 table is the d=8 metric-MDS solution of D(i,j)=B(i,i)+B(j,j)-2B(i,j) over BLOSUM62 that
 * the reference froze in hclust/src/hclust/util.hpp:21-42 (generated offline by
 * IGC/distance2coordinate/BLOSUM.m:24-30).  Row order is BLOSUM order A R N D C Q E G H I L K M F P
 * S T W Y V: row 5 = Gln (Q), row 6 = Glu (E) -- see HS_LETTER_TO_CODE (util.hpp:92).  The digits
 * are reproduced exactly; tests/test_tables.py checks them against tests/golden/constants.json
 * (dumped from the compiled reference) and against the reference's DISTANCE_SQUARE (util.hpp:43-64).
 */
#ifndef HS_TABLES_H
#define HS_TABLES_H

#define HS_ALPHABET 20   /* amino acids */
#define HS_AA_DIM 8      /* coordinates per residue (util.hpp:94 AACoordinateSize) */

static const double HS_AA_COORDS[HS_ALPHABET][HS_AA_DIM] = {
    /* Ala */ { -0.876280,   3.598596,   2.554616,  -0.729216,   0.698828,   1.221507,  -2.765205,  -3.163091},
    /* Arg */ { -4.111404,  -1.936791,  -2.682295,   0.942498,   6.924314,  -1.195785,  -1.639269,   0.615381},
    /* Asn */ { -7.471612,  -2.468058,   0.932738,  -4.488355,   0.553080,  -3.081577,   0.368010,   4.223792},
    /* Asp */ { -8.317871,  -0.848602,   1.752372,  -1.407818,  -4.874022,  -1.493568,   5.256411,  -2.561758},
    /* Cys */ {  5.421664,  11.791877,   2.675596,  -5.622478,   4.322457,   3.946839,   2.229597,  -1.901479},
    /* Gln */ { -3.771796,  -2.525005,  -1.567736,   2.619391,   2.781873,   0.952486,   3.947072,  -0.954304},
    /* Glu */ { -6.585010,  -2.752755,  -1.649014,   1.605597,  -1.833933,  -0.730211,   2.313328,  -3.239486},
    /* Gly */ { -3.978253,  -1.155062,   9.994796,  -0.195264,  -1.110059,  -2.860194,  -4.952672,  -1.495210},
    /* His */ { -2.630176,  -8.283034,  -4.773107,  -6.479084,   0.070359,   4.318067,  -1.847373,  -0.086451},
    /* Ile */ {  4.548022,   5.189698,  -3.999001,  -0.186966,  -3.275059,  -1.882387,  -0.627095,   0.049364},
    /* Leu */ {  5.341899,   4.436639,  -3.552811,   1.250614,   0.266899,  -2.609335,  -0.694939,   0.812004},
    /* Lys */ { -5.742562,  -1.207887,  -2.587323,   2.866228,   4.169821,  -1.991698,  -1.941954,  -0.747156},
    /* Met */ {  4.241223,   2.474317,  -2.658336,   2.946054,   2.011534,  -3.254331,   1.266004,  -0.186966},
    /* Phe */ {  9.340442,  -3.359172,  -0.635377,  -2.878570,  -3.255191,  -2.200202,  -1.104637,  -0.062654},
    /* Pro */ { -6.150933,   3.182318,   0.122393,   7.788554,  -3.094076,   6.831600,  -1.992627,   1.807240},
    /* Ser */ { -2.523437,   1.824168,   3.256463,  -2.386830,   0.439791,   1.024198,   0.486894,   1.190316},
    /* Thr */ { -0.823028,   3.115233,   2.075337,  -0.585875,  -1.471153,   0.518398,   1.846290,   6.269577},
    /* Trp */ { 13.592409,  -8.961858,   6.548108,   4.623650,   2.128797,   0.808588,   2.631353,   0.521535},
    /* Tyr */ {  7.173223,  -6.765800,  -2.811202,  -1.654989,  -1.878135,   3.104673,  -1.272146,  -0.635970},
    /* Val */ {  3.323480,   4.651177,  -2.996218,   1.972858,  -3.576126,  -1.427066,  -1.507041,  -0.454682},
};

/* letter - 'A' -> row of HS_AA_COORDS, -1 for B J O U X Z (util.hpp:92 base[]). */
static const signed char HS_LETTER_TO_CODE[26] = {
    /*A*/ 0, /*B*/ -1, /*C*/ 4, /*D*/ 3, /*E*/ 6, /*F*/ 13, /*G*/ 7, /*H*/ 8, /*I*/ 9,
    /*J*/ -1, /*K*/ 11, /*L*/ 10, /*M*/ 12, /*N*/ 2, /*O*/ -1, /*P*/ 14, /*Q*/ 5, /*R*/ 1,
    /*S*/ 15, /*T*/ 16, /*U*/ -1, /*V*/ 19, /*W*/ 17, /*X*/ -1, /*Y*/ 18, /*Z*/ -1};

/* row -> letter, the inverse of HS_LETTER_TO_CODE (BLOSUM order; NOT the reference's AA20 string
 * util.hpp:89, which has E and Q transposed relative to base[] -- SURVEY appendix "E/Q swap"). */
static const char HS_CODE_TO_LETTER[HS_ALPHABET + 1] = "ARNDCQEGHILKMFPSTWYV";

/* The reference's AA20 (util.hpp:89), used only by the --ref-compat-eq-swap FASTA path. */
static const char HS_REF_AA20[HS_ALPHABET + 1] = "ARNDCEQGHILKMFPSTWYV";

/* pcluster's reduced 8-class alphabet (pcluster/src/pcluster/util.hpp:100-104, REDUCEDAAINDEX):
 * [A S T] [R K E D Q] [N H] [C] [G] [I V L M] [F Y W] [P], letter - 'A' -> class, -1 for
 * B J O U X Z.  KLSH features count 3-mers over these classes: index = c0 + 8 c1 + 64 c2
 * (Kmer2Integer util.hpp:244-250 with BASEP = 1, 8, 64, ...). */
#define HS_KLSH_CLASSES 8
#define HS_KLSH_HASHLEN 3
#define HS_KLSH_FEATURES 512
static const signed char HS_REDUCED_CLASS[26] = {
    /*A*/ 0, /*B*/ -1, /*C*/ 3, /*D*/ 1, /*E*/ 1, /*F*/ 6, /*G*/ 4, /*H*/ 2, /*I*/ 5,
    /*J*/ -1, /*K*/ 1, /*L*/ 5, /*M*/ 5, /*N*/ 2, /*O*/ -1, /*P*/ 7, /*Q*/ 1, /*R*/ 1,
    /*S*/ 0, /*T*/ 0, /*U*/ -1, /*V*/ 5, /*W*/ 6, /*X*/ -1, /*Y*/ 6, /*Z*/ -1};

#endif /* HS_TABLES_H */
